#!/bin/bash
# Round 4, final tree (early exit on): 2-rank rehearsal of bench.py on one GPU (gloo), then the N-rank replays.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
(timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --rehearse-on-one-gpu > gpurun_out/r04v_rehearse2.log 2>&1; echo "exit $?" >> gpurun_out/r04v_rehearse2.log)
grep '^{"metric"' gpurun_out/r04v_rehearse2.log | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rehearsal N=2:', {k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step')}, 'transport', d['config'].get('transport'), 'errors', [k for k in d if k.endswith('_error')], 'dragon', d.get('dragon_shape', {}).get('wall_clock_to_optimum_s'), 'trimmed', d.get('trimmed_1m_outliers', {}).get('wall_clock_to_optimum_s'))"
tail -1 gpurun_out/r04v_rehearse2.log
tools/scale_replays.sh r04v \
  "X=0 8 bunny 5e-5 0.005 2" \
  "X=0 8 dragon 5e-6 0.005 1" \
  "X=0 4 dragon 5e-6 0.005 1" \
  "X=0 2 dragon 5e-6 0.005 1" \
  "X=0 4 bunny 5e-5 0.005 2" \
  "X=0 2 bunny 5e-5 0.005 2" \
  "FGOICP_REPLAY_SCHEDULE=serial 8 bunny 5e-5 0.005 2" \
  "FGOICP_REPLAY_SCHEDULE=serial 8 dragon 5e-6 0.005 1"
