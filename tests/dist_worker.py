"""World-size-N worker for tests/test_dist_gloo.py: the product's host driver (ROUND schedule) over
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
    G = np.load(os.path.join(REPO, "tests", "golden", "goicp_golden.npz"))
    d = hh.HostDriver(G[case + "tgt"], G[case + "src"], float(G[case + "res"]), float(G[case + "mse"]), schedule=1, round_width=K)
    ex = TorchExchange()
    d.set_exchange(rank, world, ex._allreduce_min, ex._allgather)
    r = d.run()
    np.savez(f"{out_prefix}.rank{rank}.npz", R=r["R"], t=r["t"], sse=r["best_sse"], exchange_calls=ex.calls,
             **{k: v for k, v in r["stats"].items()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
