cd $GRAFT_REPO_ROOT
for U in 0 4 8 0 4; do
FGOICP_UNITS_STATS=1 FGOICP_UNITS=$U python bench.py --only headline 2>/tmp/err_$U.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('units=$U', 'subcubes/s', round(d['value']), 'ms', round(d['ms_per_step'],1), 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'sse', d['result']['best_sse'])"
grep "fgoicp units" /tmp/err_$U.txt | tail -n 1
done
