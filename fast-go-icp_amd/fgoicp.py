"""Python mirror of icp::FastGoICP (fgoicp/fgoicp.hpp:8-110) over the driver-level C ABI.  The
outer/inner branch-and-bound runs in host C++ inside libfgoicp_amd.so; this class only marshals."""
import ctypes as C

import numpy as np

from . import _lib
from .nodes import from_glm
from .registration import Registration, _cloud, _fp


class FastGoICP:
    def __init__(self, pct, pcs, lut_resolution=0.005, mse_threshold=1e-3, schedule=_lib.SCHEDULE_SERIAL, round_width=1,
                 device=0, flags=0, trim_fraction=0.0):
        self._lib = _lib.load()
        pct, pcs = _cloud(pct), _cloud(pcs)
        self.nt, self.ns = len(pct), len(pcs)
        opts = _lib.SolverOpts(int(schedule), int(round_width), int(flags), int(device), float(trim_fraction))
        self._h = C.c_void_p()
        _lib.check(self._lib.fgoicp_solver_create(_fp(pct), self.nt, _fp(pcs), self.ns, float(lut_resolution), float(mse_threshold),
                                                  C.byref(opts), C.byref(self._h)), "fgoicp_solver_create")
        self._exchange = None  # keeps the ctypes callbacks alive

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fgoicp_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_exchange(self, exchange):
        """exchange: fgoicp_amd.dist.TorchExchange (or None for single process)."""
        self._exchange = exchange
        ptr = C.byref(exchange.struct) if exchange is not None else None
        _lib.check(self._lib.fgoicp_solver_set_exchange(self._h, ptr), "fgoicp_solver_set_exchange")

    def set_early_exit(self, on=True):
        """False: every subcube is evaluated in full, as the reference does (same trajectory and result; fgoicp_solver_set_early_exit)."""
        _lib.check(self._lib.fgoicp_solver_set_early_exit(self._h, int(bool(on))), "fgoicp_solver_set_early_exit")

    def run(self):
        """-> (R (3,3), t (3,)) with t restored to the callers' frame (fgoicp.cpp:29)."""
        R = np.empty(9, np.float32); t = np.empty(3, np.float32)
        _lib.check(self._lib.fgoicp_solver_run(self._h, _fp(R), _fp(t)), "fgoicp_solver_run")
        return from_glm(R), t

    def get_best_error(self):
        v = C.c_float()
        _lib.check(self._lib.fgoicp_solver_best_error(self._h, C.byref(v)), "fgoicp_solver_best_error")
        return np.float32(v.value)

    def _transform(self, fn, name):
        R = np.empty(9, np.float32); t = np.empty(3, np.float32)
        _lib.check(fn(self._h, _fp(R), _fp(t)), name)
        return from_glm(R), t

    def get_best_transform(self):
        return self._transform(self._lib.fgoicp_solver_best_transform, "fgoicp_solver_best_transform")

    def get_last_transform(self):
        return self._transform(self._lib.fgoicp_solver_last_transform, "fgoicp_solver_last_transform")

    def stats(self):
        st = _lib.RunStats()
        _lib.check(self._lib.fgoicp_solver_stats(self._h, C.byref(st)), "fgoicp_solver_stats")
        return st.as_dict()

    def preproc(self):
        offs = np.empty(6, np.float32); scale = C.c_float(); b = np.empty(6, np.float32)
        _lib.check(self._lib.fgoicp_solver_preproc(self._h, _fp(offs), C.byref(scale), _fp(b)), "fgoicp_solver_preproc")
        return dict(offset_pcs=offs[:3].copy(), offset_pct=offs[3:].copy(), scale=np.float32(scale.value), bounds=b.reshape(3, 2))

    @property
    def registration(self):
        """The operator context the solver drives (borrowed)."""
        return Registration._borrow(self._lib.fgoicp_solver_ctx(self._h), self)
