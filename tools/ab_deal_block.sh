# A/B (round 3): ROUND on N ranks, children dealt in blocks of FGOICP_DEAL_BLOCK consecutive children (siblings) instead of one by one.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_deal_block.txt
: > $OUT
for B in 1 4 8 16 1 8; do
  for WL in "bunny 5e-5 0.005 2" "dragon 5e-6 0.005 1"; do
    echo "== FGOICP_DEAL_BLOCK=$B, 8-rank replay, $WL" | tee -a $OUT
    FGOICP_DEAL_BLOCK=$B python tools/scale_replay.py 8 $WL 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  x', round(d['estimated_speedup'],2), 'balanced', round(d['ideal_if_balanced_speedup'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'mean', round(sum(d['T_rank_s'])/8*1e3,1), 'subcubes', sum(d['subcubes_rank']), 'same', d['same_optimum'])" | tee -a $OUT
  done
done
