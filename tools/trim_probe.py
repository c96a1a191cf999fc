import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fgoicp_amd as fg
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_trimming import outlier_pair
sched, K = int(sys.argv[1]), int(sys.argv[2])
tgt, src, R_gt, t_gt = outlier_pair(fg, nt=500, ns=300, frac=0.2, seed=4, angle=(100.0, 130.0))
print("creating", flush=True)
s = fg.FastGoICP(tgt, src, 0.05, 1e-3, schedule=sched, round_width=K, trim_fraction=0.25)
print("running", flush=True)
t0 = time.time(); R, t = s.run(); print("done", time.time() - t0, s.stats(), flush=True)
s.close()
