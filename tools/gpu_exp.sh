# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for M in 0 1; do
for LEG in headline; do
FGOICP_MEMO=$M timeout -k 10 300 python bench.py --only $LEG --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l)
        r=d.get('roofline')
        print('memo=$M $LEG value %.4g evaluations/launch %.0f avg_launch_us %.1f seconds_bnb %.4f' % (d['value'], r['evaluations_per_launch'], r['avg_launch_us'], d['seconds_bnb_rank0']))
"
done; done; done
FGOICP_TIMING=1 FGOICP_MEMO=1 timeout -k 10 100 python tools/run_probe.py 0 5e-5 bunny 2>&1 | grep "timing\] run"
FGOICP_TIMING=1 FGOICP_MEMO=0 timeout -k 10 100 python tools/run_probe.py 0 5e-5 bunny 2>&1 | grep "timing\] run"
