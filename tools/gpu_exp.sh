cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
rm -rf /tmp/ktrace
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktrace -- python3 $REPO/tools/dragon_probe.py 0 1e-3 synthetic1m_outliers 0.2 150 > $REPO/gpurun_out/ktrace.log 2>&1
cp /tmp/ktrace/*/*kernel_stats.csv $REPO/gpurun_out/kstats_m1.csv
