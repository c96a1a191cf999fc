cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in 0 5; do
  export FGOICP_TRIM_SAMPLE=$M FGOICP_FINALIZE_SIDE=0
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trimprof_$M -o t -- python3 $R/bench.py --only trimmed > $R/gpurun_out/trimprof_$M.log 2>&1
done
ls -R $R/gpurun_out/trimprof_0 | head
