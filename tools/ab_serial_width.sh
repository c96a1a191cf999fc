# A/B (round 3): the cap of SERIAL's speculation width (nodes of the queue evaluated ahead of the reference's pops; default 256).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_serial_width.txt
: > $OUT
for Wd in 256 512 1024 2048 256 1024; do
  echo "== FGOICP_SERIAL_WIDTH=$Wd" | tee -a $OUT
  FGOICP_SERIAL_WIDTH=$Wd python bench.py --only serial 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['serial_reference_order']
print('  serial: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes/s', round(r['subcubes_per_s']), 'subcubes', int(r['subcubes_per_step']), 'pops', r['rounds'], 'best_sse', r['best_sse'])" | tee -a $OUT
done
