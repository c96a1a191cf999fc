"""World-size-N worker for tests/test_dist_gloo.py: the product's host driver (ROUND schedule, or SERIAL with FGOICP_TEST_SCHEDULE=0) over
the oracle's operators (tests/host_harness), exchanging through fgoicp_amd.dist.TorchExchange on
the gloo backend.  Launched by torch.distributed.run; every rank writes its result."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch.distributed as dist  # noqa: E402

import fgoicp_amd as fg  # noqa: E402
from fgoicp_amd.dist import TorchExchange  # noqa: E402
from tests import host_harness as hh  # noqa: E402


def main():
    out_prefix, case, K = sys.argv[1], sys.argv[2], int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sched = int(os.environ.get("FGOICP_TEST_SCHEDULE", "1"))  # 1 = ROUND, 0 = SERIAL (the reference's order, evaluations sharded)
    if case == "kat_":  # exact copies of target points under a known motion: one unambiguous optimum (SSE = 0)
        rng = np.random.default_rng(21)
        tgt, _, _, _ = fg.synth.make_pair(400, 10, (0.156, 0.152, 0.118), seed=21)
        R_gt = fg.synth.random_rotation(rng, 150.0, 140.0)
        t_gt = np.array([0.01, -0.02, 0.015])
        src = ((tgt[:250].astype(np.float64) - t_gt) @ R_gt).astype(np.float32)
        d = hh.HostDriver(tgt, src, 0.05, 1e-3, schedule=1, round_width=K)
    else:
        G = np.load(os.path.join(REPO, "tests", "golden", "goicp_golden.npz"))
        d = hh.HostDriver(G[case + "tgt"], G[case + "src"], float(G[case + "res"]), float(G[case + "mse"]), schedule=sched, round_width=K)
    ex = TorchExchange()
    d.set_exchange(rank, world, ex._allreduce_min, ex._allgather, coop=os.environ.get("FGOICP_TEST_COOP", "0") == "1")
    r = d.run()
    np.savez(f"{out_prefix}.rank{rank}.npz", R=r["R"], t=r["t"], sse=r["best_sse"], exchange_calls=ex.calls,
             **{k: v for k, v in r["stats"].items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
