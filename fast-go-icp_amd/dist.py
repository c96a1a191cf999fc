"""Multi-GPU exchange for the sharded outer BnB: one process per GPU, torch.distributed as the
transport (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Per expansion
round the driver issues one all-reduce(MIN) of the best error and one small all-gather."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class TorchExchange:
    def __init__(self, group=None, device=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        backend = dist.get_backend(group)
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
        self.calls = 0
        self._ar = _lib.Exchange.ALLREDUCE_MIN(self._allreduce_min)
        self._ag = _lib.Exchange.ALLGATHER(self._allgather)
        self.struct = _lib.Exchange(self.rank, self.world, self._ar, self._ag, None)

    def warmup(self):
        """One all-reduce and one all-gather of a few floats: the backend's lazy communicator setup (seconds for RCCL) happens
        here instead of inside the first round of the first run."""
        a = (C.c_float * 2)(1.0, 2.0)
        r = (C.c_float * (2 * self.world))()
        return self._allreduce_min(a, 2, None) == 0 and self._allgather(a, r, 2, None) == 0

    def _allreduce_min(self, buf, n, user):
        try:
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a.copy()).to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            a[:] = t.cpu().numpy()
            self.calls += 1
            return 0
        except Exception as e:  # never let an exception cross the C ABI
            print(f"[fgoicp_amd.dist] allreduce_min failed: {e!r}", flush=True)
            return 1

    def _allgather(self, send, recv, n, user):
        try:
            s = np.ctypeslib.as_array(send, shape=(n,))
            r = np.ctypeslib.as_array(recv, shape=(n * self.world,))
            ts = torch.from_numpy(s.copy()).to(self.device)
            tr = torch.empty(n * self.world, dtype=torch.float32, device=self.device)
            dist.all_gather_into_tensor(tr, ts, group=self.group)
            r[:] = tr.cpu().numpy()
            self.calls += 1
            return 0
        except Exception as e:
            print(f"[fgoicp_amd.dist] allgather failed: {e!r}", flush=True)
            return 1
