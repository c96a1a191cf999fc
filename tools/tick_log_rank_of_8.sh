cd $GRAFT_REPO_ROOT
FGOICP_TICK_LOG=1 FGOICP_TIMING=1 python tools/scale_replay.py 8 bunny 5e-5 0.005 0 2> gpurun_out/r03_ticklog_w8.txt | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('x', round(d['estimated_speedup'],2), 'T', [round(x*1e3,1) for x in d['T_rank_s']])"
python - <<'PY'
import re, numpy as np
lines=open('gpurun_out/r03_ticklog_w8.txt').read().splitlines()
# the last 'run' block = the replay of rank 7 (repeats=0 -> one pass per rank); take ticks after the last but one '[fgoicp timing] run'
idx=[i for i,l in enumerate(lines) if l.startswith('[fgoicp timing] run ')]
print('runs logged', len(idx))
seg=lines[idx[-2]+1:idx[-1]+1] if len(idx)>=2 else lines
ev=[];us=[]
for l in seg:
    m=re.match(r'\[tick\] evals (\d+) us ([\d.]+)', l)
    if m: ev.append(int(m.group(1))); us.append(float(m.group(2)))
ev=np.array(ev); us=np.array(us)
print('last replayed rank: launches', len(ev), 'evals', ev.sum(), 'kernel ms', round(us.sum()/1e3,1))
for lo,hi in ((0,64),(64,256),(256,1024),(1024,4096),(4096,16384),(16384,10**9)):
    m=(ev>=lo)&(ev<hi)
    if m.any(): print(f'{lo:6d}-{hi:<10d} launches {m.sum():5d} evals share {ev[m].sum()/ev.sum()*100:5.1f}% time share {us[m].sum()/us.sum()*100:5.1f}%  ns/eval {us[m].sum()*1e3/ev[m].sum():7.1f}')
for l in seg:
    if 'timing] round' in l or 'timing] run' in l or 'ticks' in l: print(l[:230])
PY
