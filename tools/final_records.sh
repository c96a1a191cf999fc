# Round 3, final records: the whole GPU suite, smoke, the bench line (reads profiles/bench_pmc.json / bench_pmc_extra.json of this tree),
# a slice of the randomised whole-run campaign (SERIAL vs oracle vs SERIAL on 2-4 ranks vs ROUND) and the SERIAL 8-rank replays.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_final_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03_final_gputests.log)
tail -3 gpurun_out/r03_final_gputests.log
grep -q '^exit 0' gpurun_out/r03_final_gputests.log || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(timeout -k 10 700 python bench.py > gpurun_out/r03_final_bench_default.log 2>&1; echo "exit $?" >> gpurun_out/r03_final_bench_default.log)
grep '^{"metric"' gpurun_out/r03_final_bench_default.log > gpurun_out/profiles/r03_bench_line.json
tail -1 gpurun_out/r03_final_bench_default.log
timeout -k 10 600 python tools/fuzz_gpu.py 80 35 run 2>&1 | tail -n 2 | tee gpurun_out/r03_final_fuzz_runs.txt
: > gpurun_out/r03_final_serial_replay.jsonl
for WL in "bunny 5e-5 0.005 2" "dragon 5e-6 0.005 1"; do
  FGOICP_REPLAY_SCHEDULE=serial timeout -k 10 600 python tools/scale_replay.py 8 $WL 2>/dev/null | grep '^{' >> gpurun_out/r03_final_serial_replay.jsonl
  tail -1 gpurun_out/r03_final_serial_replay.jsonl | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['workload'], 'serial W=8 x', round(d['estimated_speedup'],2), 'with coll', round(d['estimated_speedup_with_collectives'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'host ex', d['host_exchanges_rank'][0], 'same', d['same_optimum'])"
done
