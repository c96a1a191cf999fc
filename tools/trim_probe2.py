import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fgoicp_amd as fg
from oracle import pyoracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_trimming import outlier_pair
tgt, src, R_gt, t_gt = outlier_pair(fg, nt=500, ns=300, frac=0.2, seed=4, angle=(100.0, 130.0))
print("threads", pyoracle.lib().orc_num_threads(), "cpus", os.cpu_count(), len(os.sched_getaffinity(0)), flush=True)
t0 = time.time()
o = pyoracle.FastGoICP(tgt, src, 0.05, 1e-3, trim_fraction=0.25)
print("constructed", time.time() - t0, flush=True)
r = o.run()
print("oracle done", time.time() - t0, r["stats"], flush=True)
