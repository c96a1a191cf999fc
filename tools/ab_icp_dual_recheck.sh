# Re-check (round 3, final tree): one walk for both scans of an ICP iteration (FGOICP_ICP_DUAL) below / above its adoption threshold of 262 144 points.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_icp_dual_recheck.txt
: > $OUT
leg() {
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp ms', round(r['seconds_icp_rank0']*1e3,2))"
}
for D in 0 1 0 1; do
  export FGOICP_ICP_DUAL=$D
  echo "== FGOICP_ICP_DUAL=$D" | tee -a $OUT
  leg default_threshold reference_default_threshold | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
done
