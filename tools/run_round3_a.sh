cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_multi.py tests/test_trimming.py -x -q -m gpu > gpurun_out/r3_t6.txt 2>&1; echo "pytest rc $?"; tail -n 4 gpurun_out/r3_t6.txt
bash tools/ab_layout.sh 2>&1 | tee gpurun_out/r3_ab_layout.txt
for L in 0 1; do
  FGOICP_LATE_ICP=$L python tools/scale_replay.py 8 bunny 5e-5 0.005 2 >> gpurun_out/r3_scale_replay.jsonl 2>gpurun_out/r3_scale_err.txt
  FGOICP_LATE_ICP=$L python tools/scale_replay.py 8 dragon 5e-6 0.005 1 >> gpurun_out/r3_scale_replay.jsonl 2>>gpurun_out/r3_scale_err.txt
done
python - <<PY
import json
for l in open('gpurun_out/r3_scale_replay.jsonl'):
    d=json.loads(l); print(d['workload'], 'late', d['late_icp'], 'speedup', round(d['estimated_speedup'],2), 'with coll', round(d['estimated_speedup_with_collectives'],2), 'balanced', round(d['ideal_if_balanced_speedup'],2), 'ex_us', round(d['exchange_us_rccl_world1'],1), 'icp', [round(x*1e3,1) for x in d['seconds_icp_rank']], 'T', [round(x*1e3) for x in d['T_rank_s']], d['same_optimum'])
PY
