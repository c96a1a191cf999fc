# A/B (round 3): the tail batch of ROUND (nodes a task pops per tick when its half holds few tasks) beyond 128.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_tail_batch.txt
: > $OUT
leg() {
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes', int(r.get('subcubes_per_step', d.get('subcubes_per_step',0))), 'icp ms', round(r['seconds_icp_rank0']*1e3,2), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))"
}
for B in 128 256 512 128 256; do
  export FGOICP_TAIL_BATCH=$B
  echo "== FGOICP_TAIL_BATCH=$B" | tee -a $OUT
  leg default_threshold reference_default_threshold | tee -a $OUT
  leg headline "" | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
done
for B in 128 256; do
  echo "== FGOICP_TAIL_BATCH=$B, 8-rank replay, bunny shape" | tee -a $OUT
  FGOICP_TAIL_BATCH=$B python tools/scale_replay.py 8 bunny 5e-5 0.005 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  x', round(d['estimated_speedup'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'subcubes', sum(d['subcubes_rank']))" | tee -a $OUT
done
