cd $GRAFT_REPO_ROOT
for LEG in headline dragon; do
FGOICP_TICK_LOG=1 python bench.py --only $LEG --steps 1 --warmup 1 2> gpurun_out/r03_ticklog_$LEG.txt > /dev/null
python - <<PY
import re, numpy as np
ev=[];us=[]
for l in open('gpurun_out/r03_ticklog_$LEG.txt'):
    m=re.match(r'\[tick\] evals (\d+) us ([\d.]+)', l)
    if m: ev.append(int(m.group(1))); us.append(float(m.group(2)))
n=len(ev)//2 if '$LEG'=='headline' else 0
ev=np.array(ev[n:]); us=np.array(us[n:])
print('$LEG launches', len(ev), 'evals', ev.sum(), 'kernel ms', round(us.sum()/1e3,1))
for lo,hi in ((0,64),(64,256),(256,1024),(1024,4096),(4096,16384),(16384,65536),(65536,10**9)):
    m=(ev>=lo)&(ev<hi)
    if m.any(): print(f'{lo:6d}-{hi:<10d} launches {m.sum():5d} evals share {ev[m].sum()/ev.sum()*100:5.1f}% time share {us[m].sum()/us.sum()*100:5.1f}%  ns/eval {us[m].sum()*1e3/ev[m].sum():7.1f}')
PY
done
