# kernel timeline of tools/icp_bench.py for one loop variant:  bash tools/prof_icp.sh <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
export ICP_VARIANT=$TAG
rm -rf /tmp/icp_tr_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icp_tr_$TAG -- python3 $GRAFT_REPO_ROOT/tools/icp_bench.py bunny 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_icp_prof_$TAG.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/trace_dump.py /tmp/icp_tr_$TAG 60 > $GRAFT_REPO_ROOT/gpurun_out/r3_icp_timeline_$TAG.txt
cp /tmp/icp_tr_$TAG/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r3_icp_kernel_stats_$TAG.csv
