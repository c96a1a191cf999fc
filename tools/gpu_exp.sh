# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for Z in 2 3; do
echo "layout $Z: $(FGOICP_LUT_ZPAIR=$Z timeout -k 10 300 python tools/op_bench.py bunny 1024 5 2>/dev/null)"
done; done
for Z in 2 3; do
FGOICP_LUT_ZPAIR=$Z timeout -k 10 200 python bench.py --only headline --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); r=d['roofline']; print('layout $Z headline value %.4g kernel GB/s %.0f avg_launch_us %.1f best_sse %r' % (d['value'], r['achieved'], r['avg_launch_us'], d['result']['best_sse']))
"
done
