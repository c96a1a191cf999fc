"""Work split of the sharded search: runs tests/gpu_dist_worker.py with W ranks on ONE GPU (gloo exchange) and prints every
rank's share.  Wall times are meaningless here (the ranks share the device); the subcube counts are what N GPUs would see.

    python tools/dist_balance.py <world> [workload] [mse] [round_width]
"""
import os, subprocess, sys, tempfile, socket
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
world = int(sys.argv[1]); workload = sys.argv[2] if len(sys.argv) > 2 else "bunny"; mse = sys.argv[3] if len(sys.argv) > 3 else "5e-5"; K = sys.argv[4] if len(sys.argv) > 4 else "0"
with socket.socket() as so:
    so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
d = tempfile.mkdtemp()
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
       os.path.join(REPO, "tests", "gpu_dist_worker.py"), os.path.join(d, "w"), workload, mse, K]
p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
if p.returncode:
    print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
rs = [np.load(os.path.join(d, f"w.rank{r}.npz")) for r in range(world)]
tc = np.array([int(r["trans_cubes"]) for r in rs]); rc = [int(r["rot_cubes"]) for r in rs]
print(f"world {world} {workload} mse {mse} K {K}: rounds {int(rs[0]['rounds'])}, sse {float(rs[0]['sse']):.6g}")
print(" subcubes per rank", tc.tolist(), " total", int(tc.sum()), " max/mean", float(tc.max() / tc.mean()))
print(" rot cubes per rank", rc, " icp runs", [int(r["icp_runs"]) for r in rs], " icp s", [round(float(r["seconds_icp"]), 3) for r in rs], " seconds", [round(float(r["seconds"]), 2) for r in rs])
