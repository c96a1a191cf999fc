cd $GRAFT_REPO_ROOT
bash tools/pmc_extra.sh r03 "1 2 4" headline dragon trimmed > gpurun_out/r03_pmc_extra.log 2>&1
tail -3 gpurun_out/r03_pmc_extra.log | cut -c1-200
FGOICP_TIMING=1 python bench.py --only default_threshold --steps 1 --warmup 1 2> gpurun_out/r03_timing_default_threshold.txt > /dev/null
grep -c 'timing' gpurun_out/r03_timing_default_threshold.txt
