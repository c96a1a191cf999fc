"""Strong-scaling estimate on ONE GPU.  A W-rank run (all ranks on device 0, exchange on gloo) records what every exchange
returned; then each rank's share is replayed ALONE on the GPU against the recorded exchange results, which times what that
rank would do on its own MI355X (everything but the RCCL latency itself: two small collectives per round).

    python tools/scale_replay.py <world> [workload] [mse] [res]      -> JSON line with T(1), max_r T_r(W) and the ratio
"""
import ctypes as C
import json
import os
import pickle
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def record_worker(prefix, workload, mse, res):
    import torch.distributed as dist
    import fgoicp_amd as fg
    from fgoicp_amd.dist import TorchExchange

    class Recording(TorchExchange):
        def __init__(self):
            super().__init__()
            self.log = []
            self._ar = fg._lib.Exchange.ALLREDUCE_MIN(self._rec_ar)
            self._ag = fg._lib.Exchange.ALLGATHER(self._rec_ag)
            self.struct = fg._lib.Exchange(self.rank, self.world, self._ar, self._ag, None)

        def _rec_ar(self, buf, n, user):
            rc = self._allreduce_min(buf, n, user)
            self.log.append(("ar", np.ctypeslib.as_array(buf, shape=(n,)).copy()))
            return rc

        def _rec_ag(self, send, recv, n, user):
            rc = self._allgather(send, recv, n, user)
            self.log.append(("ag", np.ctypeslib.as_array(recv, shape=(n * self.world,)).copy()))
            return rc

    dist.init_process_group("gloo")
    tgt, src, _, _ = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=0, device=0)
    ex = Recording()
    s.set_exchange(ex)
    s.run()
    pickle.dump(ex.log, open(f"{prefix}.rank{ex.rank}.pkl", "wb"))
    dist.barrier()
    s.close()
    dist.destroy_process_group()


def replay(prefix, rank, world, workload, mse, res, repeats=3):
    import fgoicp_amd as fg
    log = pickle.load(open(f"{prefix}.rank{rank}.pkl", "rb"))
    state = {"i": 0}

    def ar(buf, n, user):
        kind, data = log[state["i"]]; state["i"] += 1
        np.ctypeslib.as_array(buf, shape=(n,))[:] = data
        return 0

    def ag(send, recv, n, user):
        kind, data = log[state["i"]]; state["i"] += 1
        np.ctypeslib.as_array(recv, shape=(n * world,))[:] = data
        return 0

    ar_c, ag_c = fg._lib.Exchange.ALLREDUCE_MIN(ar), fg._lib.Exchange.ALLGATHER(ag)

    class Ex:
        struct = fg._lib.Exchange(rank, world, ar_c, ag_c, None)

    tgt, src, _, _ = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=0, device=0)
    if world > 1:
        s.set_exchange(Ex())
    best = 1e30
    for _ in range(repeats + 1):  # first pass warms up
        state["i"] = 0
        t0 = time.perf_counter()
        s.run()
        best = min(best, time.perf_counter() - t0)
    st = s.stats()
    s.close()
    return best, st


def main():
    if sys.argv[1] == "--record":
        record_worker(sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5]))
        return
    world = int(sys.argv[1]); workload = sys.argv[2] if len(sys.argv) > 2 else "bunny"
    mse = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5; res = float(sys.argv[4]) if len(sys.argv) > 4 else 0.005
    d = tempfile.mkdtemp()
    prefix = os.path.join(d, "rec")
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), "--record", prefix, workload, repr(mse), repr(res)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if p.returncode:
        print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
    t1, st1 = replay(prefix, 0, 1, workload, mse, res) if False else (None, None)
    import fgoicp_amd as fg  # single-rank reference time
    tgt, src, _, _ = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=0, device=0)
    t1 = 1e30
    for _ in range(3):
        t0 = time.perf_counter(); s.run(); t1 = min(t1, time.perf_counter() - t0)
    sub1 = s.stats()["trans_cubes"]
    s.close()
    times, subs = [], []
    only = os.environ.get("REPLAY_ONLY_RANK")
    for r in ([int(only)] if only else range(world)):
        t, st = replay(prefix, r, world, workload, mse, res, repeats=1 if only else 3)
        times.append(t); subs.append(int(st["trans_cubes"]))
    print(json.dumps({"workload": workload, "world": world, "T1_s": t1, "subcubes_1": int(sub1), "T_rank_s": times, "subcubes_rank": subs,
                      "estimated_speedup": t1 / max(times), "estimated_efficiency": t1 / max(times) / world,
                      "note": "each rank's share replayed alone on one GPU against the recorded exchange results; RCCL latency (2 small collectives per round) not included"}))


if __name__ == "__main__":
    main()
