# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=6 2>&1 | tail -14
timeout -k 10 300 python bench.py --only default_threshold 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); r=d['reference_default_threshold']; print({k:r[k] for k in ('wall_clock_to_optimum_s','seconds_icp_rank0','icp_runs_rank0','subcubes_per_step','best_sse')})
"
for W in 8; do timeout -k 10 300 python tools/scale_replay.py $W bunny 5e-5 0.005 2 2>/dev/null | cut -c1-700; done
timeout -k 10 400 python tools/scale_replay.py 8 dragon 5e-6 0.005 1 2>/dev/null | cut -c1-700
