# A/B (round 3): how SERIAL's speculation width starts and grows (FGOICP_SERIAL_START / FGOICP_SERIAL_GROW); the trajectory is the same by construction.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_serial_growth.txt
: > $OUT
for CFG in "1 2" "1 4" "2 4" "4 4" "8 4" "4 8" "16 4" "1 2" "4 4"; do
  set -- $CFG
  echo "== FGOICP_SERIAL_START=$1 FGOICP_SERIAL_GROW=$2" | tee -a $OUT
  FGOICP_SERIAL_START=$1 FGOICP_SERIAL_GROW=$2 python bench.py --only serial 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['serial_reference_order']
print('  serial: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes/s', round(r['subcubes_per_s']), 'subcubes', int(r['subcubes_per_step']), 'pops', r['rounds'], 'best_sse', r['best_sse'])" | tee -a $OUT
done
