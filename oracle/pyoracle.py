"""ctypes binding of the CPU oracle (oracle/libgoicp_oracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); the product
package never imports this module.  PARITY UNPINNED — see oracle/goicp_oracle.hpp."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "libgoicp_oracle.so")
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in ("goicp_oracle.cpp", "capi.cpp", "goicp_oracle.hpp", "Makefile")]
    if force or not os.path.exists(_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_PATH) for s in srcs):
        subprocess.run(["make", "-C", _DIR, "-B" if force else "-s"], check=True)
    return _PATH


_lib = None


def usable_cpus(cap=16):
    """CPUs this process may really use: affinity mask, cgroup CPU quota, and at most `cap` (a GPU box shows every core of
    the host but grants a share of ~16; 256 OpenMP threads spinning on a 16-CPU quota turn a 0.4 s run into minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = C.CDLL(_PATH)
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        want = os.environ.get("FGOICP_ORACLE_THREADS") or os.environ.get("OMP_NUM_THREADS")
        L.orc_set_num_threads(int(want) if want and want.isdigit() else usable_cpus())
        L.orc_convention.argtypes = [C.c_char_p, C.c_int, _ip]
        L.orc_convention.restype = C.c_int
        L.orc_rotation.argtypes = [C.c_float, C.c_float, C.c_float, _fp, _fp, _ip]
        L.orc_rotnode_overlaps.argtypes = [C.c_float] * 4
        L.orc_rotnode_overlaps.restype = C.c_int
        L.orc_transnode_pop_order.argtypes = [_fp, _fp, C.c_int, _ip]
        L.orc_reg_create.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_float, C.c_int, C.c_int]
        L.orc_reg_create.restype = C.c_void_p
        L.orc_reg_destroy.argtypes = [C.c_void_p]
        L.orc_reg_set_inliers.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_reg_use_grid.argtypes = [C.c_void_p, C.c_int]
        L.orc_goicp_create_trim.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, C.c_float]
        L.orc_goicp_create_trim.restype = C.c_void_p
        L.orc_reg_lut_dims.argtypes = [C.c_void_p, _ip]
        L.orc_reg_lut_get.argtypes = [C.c_void_p, _fp]
        L.orc_reg_lut_set.argtypes = [C.c_void_p, _fp]
        L.orc_reg_lut_search.argtypes = [C.c_void_p, _fp, C.c_size_t, _fp]
        L.orc_reg_bounds.argtypes = [C.c_void_p, _fp, C.c_float, _fp, C.c_int, C.c_int, _fp, _fp]
        L.orc_reg_sse.argtypes = [C.c_void_p, _fp, _fp]
        L.orc_reg_sse.restype = C.c_float
        L.orc_reg_icp.argtypes = [C.c_void_p, _fp, _fp, C.c_size_t, C.c_float, _fp, _fp, _fp, _ip]
        L.orc_reg_procrustes.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _ip]
        L.orc_closest_orthogonal.argtypes = [_fp, _fp]
        L.orc_svd3.argtypes = [_dp, _dp, _dp, _dp]
        L.orc_goicp_create.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float]
        L.orc_goicp_create.restype = C.c_void_p
        L.orc_goicp_destroy.argtypes = [C.c_void_p]
        L.orc_goicp_use_grid.argtypes = [C.c_void_p, C.c_int]
        L.orc_goicp_preproc.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, _fp, _ip]
        L.orc_goicp_run.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, C.POINTER(C.c_ulonglong)]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(_fp)


def to_glm(R):
    return np.ascontiguousarray(np.asarray(R, dtype=np.float32).T).reshape(9)


def from_glm(flat):
    return np.asarray(flat, dtype=np.float32).reshape(3, 3).T.copy()


CONVENTION_DEFAULTS = dict(fma_matvec=1, fma_dist=1, fma_rot_sub=0, fma_trans_sub=0, tex_weight=0, tex_blend=0, sum_mode=0, sin_ulps=0,
                           svd_r2_two_sided=0)


def set_conventions(**kw):
    """Flip conventions the oracle CHOOSES (goicp_oracle.hpp: Conventions); process-global.  Returns the previous values."""
    old = {}
    for name, value in kw.items():
        prev = C.c_int()
        if lib().orc_convention(name.encode(), int(value), C.byref(prev)) != 0:
            raise KeyError(name)
        old[name] = prev.value
    return old


def get_conventions():
    out = {}
    for name in CONVENTION_DEFAULTS:
        prev = C.c_int()
        lib().orc_convention(name.encode(), -1000, C.byref(prev))
        out[name] = prev.value
    return out


def reset_conventions():
    lib().orc_conventions_reset()


def rotation(x, y, z):
    R = np.empty(9, np.float32); r = C.c_float(); ok = C.c_int()
    lib().orc_rotation(x, y, z, _f(R), C.byref(r), C.byref(ok))
    return from_glm(R), np.float32(r.value), bool(ok.value)


def rotnode_overlaps(x, y, z, span):
    return bool(lib().orc_rotnode_overlaps(x, y, z, span))


def transnode_pop_order(lb, span):
    lb = np.ascontiguousarray(lb, np.float32); span = np.ascontiguousarray(span, np.float32)
    out = np.empty(len(lb), np.int32)
    lib().orc_transnode_pop_order(_f(lb), _f(span), len(lb), out.ctypes.data_as(_ip))
    return out


def closest_orthogonal(ABt_glm9):
    a = np.ascontiguousarray(ABt_glm9, np.float32).reshape(9)
    out = np.empty(9, np.float32)
    lib().orc_closest_orthogonal(_f(a), _f(out))
    return out


def svd3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    U = np.empty(9); S = np.empty(3); V = np.empty(9)
    lib().orc_svd3(A.ctypes.data_as(_dp), U.ctypes.data_as(_dp), S.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return U.reshape(3, 3), S, V.reshape(3, 3)


class Registration:
    """goicp_oracle::Registration (restates fgoicp/registration.hpp:49-98)."""

    def __init__(self, pct, pcs, bounds, lut_resolution, build_lut=True, quantize=True):
        self.pct = np.ascontiguousarray(pct, np.float32); self.pcs = np.ascontiguousarray(pcs, np.float32)
        self.nt, self.ns = len(self.pct), len(self.pcs)
        b = np.asarray(bounds, np.float32).reshape(6)
        self._h = C.c_void_p(lib().orc_reg_create(_f(self.pct), self.nt, _f(self.pcs), self.ns, _f(b), lut_resolution, int(build_lut), int(quantize)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_reg_destroy(self._h)
            self._h = None

    def set_inliers(self, k):
        """EXTENSION: trimmed sums over the k smallest per-point terms (0 = off)."""
        lib().orc_reg_set_inliers(self._h, int(k))

    def use_grid(self, on=True):
        """CPU baseline only: exact nearest neighbours through a uniform grid instead of the O(n*m) loops (same results)."""
        lib().orc_reg_use_grid(self._h, int(bool(on)))

    def lut_dims(self):
        d = (C.c_int * 3)()
        lib().orc_reg_lut_dims(self._h, d)
        return tuple(d)

    def lut_get(self):
        dx, dy, dz = self.lut_dims()
        out = np.empty(dx * dy * dz, np.float32)
        lib().orc_reg_lut_get(self._h, _f(out))
        return out.reshape(dz, dy, dx)

    def lut_set(self, data):
        a = np.ascontiguousarray(data, np.float32).reshape(-1)
        dx, dy, dz = self.lut_dims()
        assert a.size == dx * dy * dz
        lib().orc_reg_lut_set(self._h, _f(a))

    def lut_search(self, q):
        q = np.ascontiguousarray(q, np.float32)
        out = np.empty(len(q), np.float32)
        lib().orc_reg_lut_search(self._h, _f(q), len(q), _f(out))
        return out

    def compute_bounds(self, R, rot_span, tnodes4, fix_rot):
        tn = np.ascontiguousarray(tnodes4, np.float32).reshape(-1, 4)
        lb = np.empty(len(tn), np.float32); ub = np.empty(len(tn), np.float32)
        lib().orc_reg_bounds(self._h, _f(to_glm(R)), rot_span, _f(tn), len(tn), int(bool(fix_rot)), _f(lb), _f(ub))
        return lb, ub

    def compute_sse_error(self, R, t):
        t = np.ascontiguousarray(t, np.float32)
        return np.float32(lib().orc_reg_sse(self._h, _f(to_glm(R)), _f(t)))

    def icp(self, R, t, max_iter, thr):
        t = np.ascontiguousarray(t, np.float32)
        sse = C.c_float(); it = C.c_int(); Ro = np.empty(9, np.float32); to = np.empty(3, np.float32)
        lib().orc_reg_icp(self._h, _f(to_glm(R)), _f(t), max_iter, thr, C.byref(sse), _f(Ro), _f(to), C.byref(it))
        return np.float32(sse.value), from_glm(Ro), to, it.value

    def procrustes(self, working):
        w = np.ascontiguousarray(working, np.float32)
        R = np.empty(9, np.float32); t = np.empty(3, np.float32); cen = np.empty(6, np.float32); ABt = np.empty(9, np.float32)
        idx = np.empty(self.ns, np.int32)
        lib().orc_reg_procrustes(self._h, _f(w), _f(R), _f(t), _f(cen), _f(ABt), idx.ctypes.data_as(_ip))
        return from_glm(R), t, cen, ABt, idx


class FastGoICP:
    """goicp_oracle::FastGoICP (restates fgoicp/fgoicp.hpp, fgoicp.cpp)."""

    def __init__(self, pct, pcs, lut_resolution, mse_threshold, trim_fraction=0.0):
        pct = np.ascontiguousarray(pct, np.float32); pcs = np.ascontiguousarray(pcs, np.float32)
        self.nt, self.ns = len(pct), len(pcs)
        self._h = C.c_void_p(lib().orc_goicp_create_trim(_f(pct), self.nt, _f(pcs), self.ns, lut_resolution, mse_threshold, trim_fraction))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_goicp_destroy(self._h)
            self._h = None

    def use_grid(self, on=True):
        """exact nearest neighbours through the uniform grid instead of the O(ns*nt) loops (same results, bit for bit)"""
        lib().orc_goicp_use_grid(self._h, int(bool(on)))

    def preproc(self):
        offs = np.empty(6, np.float32); scale = C.c_float(); bounds = np.empty(6, np.float32)
        t = np.empty((self.nt, 3), np.float32); s = np.empty((self.ns, 3), np.float32); dims = (C.c_int * 3)()
        lib().orc_goicp_preproc(self._h, _f(offs), C.byref(scale), _f(bounds), _f(t), _f(s), dims)
        return dict(offset_pcs=offs[:3].copy(), offset_pct=offs[3:].copy(), scale=np.float32(scale.value), bounds=bounds.reshape(3, 2),
                    pct=t, pcs=s, lut_dims=tuple(dims))

    def run(self):
        R = np.empty(9, np.float32); t = np.empty(3, np.float32); sse = C.c_float(); ts = np.empty(3, np.float32)
        st = (C.c_ulonglong * 7)()
        lib().orc_goicp_run(self._h, _f(R), _f(t), C.byref(sse), _f(ts), st)
        names = ["trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb"]
        return dict(R=from_glm(R), t=t, best_sse=np.float32(sse.value), t_scaled=ts, stats={n: int(st[i]) for i, n in enumerate(names)})
