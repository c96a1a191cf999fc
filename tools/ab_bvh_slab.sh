# A/B (round 3): leaf slabs of the target tree (FGOICP_BVH_SLAB: a leaf's points between two parallel planes; the per-query leaf test takes max(box, slab) distance).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_bvh_slab.txt
: > $OUT
leg() {
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
il=r.get('icp_latency') or d.get('icp_latency') or {}
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp ms', round(r['seconds_icp_rank0']*1e3,2), 'us/iter', round(il.get('us_per_iteration',0),1), 'setup s', round(r['setup_s_upload_plus_lut_build'],3), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))"
}
for I in 0 1 0 1; do
  export FGOICP_BVH_SLAB=$I
  echo "== FGOICP_BVH_SLAB=$I" | tee -a $OUT
  for W in bunny dragon; do
    ICP_VARIANT=default python tools/icp_bench.py $W 5 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  icp_bench', d['workload'], 'thr', d['thr'], 'iters', d['iters'], 'us/iter', round(d['us_per_iter'],1), 'sse', d['sse'])" | tee -a $OUT
  done
  leg default_threshold reference_default_threshold | tee -a $OUT
  leg headline "" | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
  leg trimmed trimmed_1m_outliers | tee -a $OUT
done
