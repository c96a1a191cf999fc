"""World-size-N worker for tests/test_gpu_cli_dist.py: the HIP solver (ROUND schedule, sharded over the ranks) with the
exchange on the gloo backend, every rank on the SAME GPU (a one-GPU box cannot host two RCCL ranks).  Launched by
torch.distributed.run; every rank writes its result and its share of the work."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch.distributed as dist  # noqa: E402

import fgoicp_amd as fg  # noqa: E402
from fgoicp_amd.dist import TorchExchange  # noqa: E402


def main():
    out_prefix, workload, mse, K = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    tgt, src, R_gt, t_gt = fg.synth.workload(workload, angle_deg=150.0, min_angle_deg=110.0)
    s = fg.FastGoICP(tgt, src, 0.005 if workload in ("bunny", "dragon") else 0.02, mse, schedule=fg.SCHEDULE_ROUND, round_width=K, device=0)
    ex = TorchExchange()
    s.set_exchange(ex)
    dist.barrier()
    t0 = time.perf_counter()
    R, t = s.run()
    sec = time.perf_counter() - t0
    st = s.stats()
    np.savez(f"{out_prefix}.rank{rank}.npz", R=R, t=t, sse=float(s.get_best_error()), seconds=sec, exchange_calls=ex.calls, **{k: v for k, v in st.items()})
    dist.barrier()
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
