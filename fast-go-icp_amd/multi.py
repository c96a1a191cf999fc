"""Multi-GPU inside libfgoicp_amd.so (include/fgoicp_amd.h, "Multi-GPU inside the library"): the RCCL transport of the exchange
hook and the one-process / one-thread-per-GPU runner.  ctypes marshalling only."""
import ctypes as C

import numpy as np

from . import _lib
from .nodes import from_glm, to_glm
from .registration import Registration, _cloud, _fp


def rccl_unique_id():
    """128-byte ncclUniqueId (bytes): draw it on rank 0 and hand it to the other ranks."""
    buf = (C.c_ubyte * 128)()
    _lib.check(_lib.load().fgoicp_rccl_unique_id(buf), "fgoicp_rccl_unique_id")
    return bytes(buf)


class RcclExchange:
    """One rank's RCCL communicator as an fgoicp_exchange (ncclCommInitRank blocks until every rank has called it).
    Drop-in for fgoicp_amd.dist.TorchExchange in FastGoICP.set_exchange: no Python in the per-round collectives."""

    def __init__(self, rank, world, unique_id, device, nonblocking=False):
        self._lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        ident = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        self._h = C.c_void_p()
        _lib.check(self._lib.fgoicp_rccl_create_ex(self.rank, self.world, ident, int(device), int(bool(nonblocking)), C.byref(self._h)), "fgoicp_rccl_create_ex")
        self.struct = _lib.Exchange()
        _lib.check(self._lib.fgoicp_rccl_exchange(self._h, C.byref(self.struct)), "fgoicp_rccl_exchange")

    @property
    def calls(self):
        n = C.c_uint64()
        _lib.check(self._lib.fgoicp_rccl_calls(self._h, C.byref(n)), "fgoicp_rccl_calls")
        return n.value

    @property
    def comm_count(self):
        """ranks the RCCL communicator itself reports (ncclCommCount)"""
        n = C.c_int()
        _lib.check(self._lib.fgoicp_rccl_comm_count(self._h, C.byref(n)), "fgoicp_rccl_comm_count")
        return n.value

    def test_inprogress(self, n):
        """TEST HOOK: the next n collectives report ncclInProgress once; returns the polls of ncclCommGetAsyncError made so far."""
        polls = C.c_uint64()
        _lib.check(self._lib.fgoicp_rccl_test_inprogress(self._h, int(n), C.byref(polls)), "fgoicp_rccl_test_inprogress")
        return polls.value

    def abort(self):
        """A peer rank failed: the collective in flight here and every later one must end (fgoicp_rccl_abort raises the flag; the
        thread running the collectives aborts the communicator)."""
        _lib.check(self._lib.fgoicp_rccl_abort(self._h), "fgoicp_rccl_abort")

    def warmup(self):
        a = (C.c_float * 2)(1.0, 2.0)
        r = (C.c_float * (2 * self.world))()
        return self.struct.allreduce_min(a, 2, self.struct.user) == 0 and self.struct.allgather(a, r, 2, self.struct.user) == 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fgoicp_rccl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGoICP:
    """icp::FastGoICP sharded over several GPUs of one node from ONE process (one host thread + one solver per device)."""

    def __init__(self, pct, pcs, lut_resolution=0.005, mse_threshold=1e-3, devices=(0,), transport=_lib.TRANSPORT_RCCL, round_width=0, flags=0,
                 trim_fraction=0.0, schedule=_lib.SCHEDULE_ROUND):
        self._lib = _lib.load()
        pct, pcs = _cloud(pct), _cloud(pcs)
        opts = _lib.SolverOpts(int(schedule), int(round_width), int(flags), 0, float(trim_fraction))  # SCHEDULE_SERIAL: the reference's trajectory, sharded
        dev = np.asarray(list(devices), np.int32)
        self._h = C.c_void_p()
        _lib.check(self._lib.fgoicp_multi_create(_fp(pct), len(pct), _fp(pcs), len(pcs), float(lut_resolution), float(mse_threshold), C.byref(opts),
                                                 dev.ctypes.data_as(_lib.c_int_p), len(dev), int(transport), C.byref(self._h)), "fgoicp_multi_create")
        self.world = len(dev)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fgoicp_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self):
        R = np.empty(9, np.float32); t = np.empty(3, np.float32)
        _lib.check(self._lib.fgoicp_multi_run(self._h, _fp(R), _fp(t)), "fgoicp_multi_run")
        return from_glm(R), t

    def icp(self, R0, t0, max_iter=100, convergence_threshold=0.005):
        """One ICP run executed by all ranks together (fgoicp_multi_icp) -> (sse, R, t, iterations); in the solver's scaled frame."""
        sse = np.empty(1, np.float32); R = np.empty(9, np.float32); t = np.empty(3, np.float32); it = np.empty(1, np.int32)
        _lib.check(self._lib.fgoicp_multi_icp(self._h, _fp(to_glm(np.asarray(R0, np.float32))), _fp(np.ascontiguousarray(np.asarray(t0, np.float32))), int(max_iter),
                                              float(convergence_threshold), _fp(sse), _fp(R), _fp(t), it.ctypes.data_as(_lib.c_int_p)), "fgoicp_multi_icp")
        return np.float32(sse[0]), from_glm(R), t, int(it[0])

    def set_early_exit(self, on=True):
        for r in range(self.world):
            _lib.check(self._lib.fgoicp_solver_set_early_exit(C.c_void_p(self._lib.fgoicp_multi_solver(self._h, r)), int(bool(on))), "fgoicp_solver_set_early_exit")

    def test_fault(self, rank, call):
        """TEST HOOK: the call-th exchange of `rank` in the next run fails, once."""
        _lib.check(self._lib.fgoicp_multi_test_fault(self._h, int(rank), int(call)), "fgoicp_multi_test_fault")

    def set_record(self, on=True):
        _lib.check(self._lib.fgoicp_multi_set_record(self._h, int(bool(on))), "fgoicp_multi_set_record")

    def recorded(self, rank=0):
        """(host-side collectives of `rank`, device all-gathers) of the last recorded run"""
        h, d = C.c_uint64(), C.c_uint64()
        _lib.check(self._lib.fgoicp_multi_recorded(self._h, int(rank), C.byref(h), C.byref(d)), "fgoicp_multi_recorded")
        return h.value, d.value

    def replay_rank(self, rank):
        """Wall-clock of `rank` running alone against the recorded exchange results."""
        s = C.c_double()
        _lib.check(self._lib.fgoicp_multi_replay_rank(self._h, int(rank), C.byref(s)), "fgoicp_multi_replay_rank")
        return s.value

    def seconds(self, rank):
        s = C.c_double()
        _lib.check(self._lib.fgoicp_multi_seconds(self._h, int(rank), C.byref(s)), "fgoicp_multi_seconds")
        return s.value

    def stats(self, rank=0):
        st = _lib.RunStats()
        _lib.check(self._lib.fgoicp_solver_stats(C.c_void_p(self._lib.fgoicp_multi_solver(self._h, int(rank))), C.byref(st)), "fgoicp_solver_stats")
        return st.as_dict()

    def get_best_error(self, rank=0):
        v = C.c_float()
        _lib.check(self._lib.fgoicp_solver_best_error(C.c_void_p(self._lib.fgoicp_multi_solver(self._h, int(rank))), C.byref(v)), "fgoicp_solver_best_error")
        return np.float32(v.value)

    def registration(self, rank=0):
        s = C.c_void_p(self._lib.fgoicp_multi_solver(self._h, int(rank)))
        return Registration._borrow(self._lib.fgoicp_solver_ctx(s), self)
