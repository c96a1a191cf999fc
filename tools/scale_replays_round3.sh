# Round 3, session 3: the cooperative ICP around the dual walk (tests), then the 8-rank replays under the final tree (k-d orders).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/r03_final_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03_final_gputests.log)
tail -3 gpurun_out/r03_final_gputests.log
grep -q '^exit 0' gpurun_out/r03_final_gputests.log || exit 1
: > gpurun_out/r03_final_scale_replay.jsonl
run() {  # env-assignments... then 5 args
  echo "== $*"
  env "${@:1:$#-5}" timeout -k 10 900 python tools/scale_replay.py "${@: -5}" 2> gpurun_out/r03_final_replay_err.log | grep '^{' >> gpurun_out/r03_final_scale_replay.jsonl || { tail -5 gpurun_out/r03_final_replay_err.log; return 1; }
  tail -1 gpurun_out/r03_final_scale_replay.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['workload'], d['schedule'], d['world'], 'coop', d['coop_icp'], 'T1', round(d['T1_s'],3), 'x', round(d['estimated_speedup'],2), 'with coll', round(d['estimated_speedup_with_collectives'],2), 'balanced', round(d['ideal_if_balanced_speedup'],2), 'icp', [round(x*1e3,1) for x in d['seconds_icp_rank']], 'T', [round(x*1e3,1) for x in d['T_rank_s']], 'same', d['same_optimum'], 'host ex', d['host_exchanges_rank'][0], 'dev gathers', d['device_allgathers'])"
}
run X=0 8 bunny 5e-5 0.005 2 &&
run FGOICP_COOP_ICP=1 8 bunny 5e-5 0.005 2 &&
run X=0 8 dragon 5e-6 0.005 1 &&
run FGOICP_COOP_SPLIT_MIN=0 8 dragon 5e-6 0.005 1 &&
run FGOICP_COOP_ICP=0 8 dragon 5e-6 0.005 1 &&
run X=0 4 dragon 5e-6 0.005 1 &&
run X=0 2 dragon 5e-6 0.005 1 &&
run X=0 4 bunny 5e-5 0.005 2 &&
run X=0 2 bunny 5e-5 0.005 2 &&
run FGOICP_REPLAY_SCHEDULE=serial 8 bunny 5e-5 0.005 2 &&
run FGOICP_REPLAY_SCHEDULE=serial 8 dragon 5e-6 0.005 1 &&
run FGOICP_REPLAY_TRIM=0.2 FGOICP_COOP_ICP=0 8 synthetic1m_outliers 1e-3 0.005 0 &&
run FGOICP_REPLAY_TRIM=0.2 FGOICP_COOP_ICP=1 8 synthetic1m_outliers 1e-3 0.005 0
