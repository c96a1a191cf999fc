cd $GRAFT_REPO_ROOT
for Wd in 256 0; do
  for WL in "bunny 5e-5 0.005 2" "dragon 5e-6 0.005 1"; do
    if [ $Wd = 0 ]; then unset FGOICP_SERIAL_WIDTH; else export FGOICP_SERIAL_WIDTH=$Wd; fi
    FGOICP_REPLAY_SCHEDULE=serial python tools/scale_replay.py 8 $WL 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('width', '$Wd', d['workload'], 'x', round(d['estimated_speedup'],2), 'with coll', round(d['estimated_speedup_with_collectives'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'host ex', d['host_exchanges_rank'][0], 'same', d['same_optimum'])"
  done
done
