"""Builds and binds tests/host_harness/harness.cpp (driver template x oracle operators). Test-only."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(os.path.dirname(_DIR))
_SO = os.path.join(_DIR, "libhost_harness.so")
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
AR = C.CFUNCTYPE(C.c_int, _fp, C.c_size_t, C.c_void_p)
AG = C.CFUNCTYPE(C.c_int, _fp, _fp, C.c_size_t, C.c_void_p)


def build():
    from oracle import pyoracle
    pyoracle.build()
    deps = [os.path.join(_DIR, "harness.cpp"), os.path.join(_REPO, "fast-go-icp_amd/csrc/host/driver.hpp"), os.path.join(_REPO, "fast-go-icp_amd/csrc/device/morton.hpp"),
            os.path.join(_REPO, "fast-go-icp_amd/csrc/host/math3.hpp"), os.path.join(_REPO, "oracle/libgoicp_oracle.so"),
            os.path.join(_DIR, "oracle_ops.hpp"), os.path.join(_REPO, "fast-go-icp_amd/csrc/device/slab.hpp"), os.path.join(_REPO, "fast-go-icp_amd/csrc/host/knobs.hpp")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        tmp = f"{_SO}.{os.getpid()}.tmp"  # several ranks of a world-size-N test may get here together: build aside, rename atomically
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DFGOICP_DEV_KNOBS", "-fopenmp", "-shared", "-o", tmp,
                        os.path.join(_DIR, "harness.cpp"), "-L" + os.path.join(_REPO, "oracle"), "-lgoicp_oracle",
                        "-Wl,-rpath," + os.path.join(_REPO, "oracle")], check=True)
        os.replace(tmp, _SO)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.harness_create.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int]
        L.harness_create.restype = C.c_void_p
        L.harness_create_trim.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]
        L.harness_create_trim.restype = C.c_void_p
        L.harness_create_ex.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]
        L.harness_create_ex.restype = C.c_void_p
        L.harness_lut_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.harness_lut_set.argtypes = [C.c_void_p, _fp, C.c_size_t]
        L.harness_seconds.argtypes = [C.c_void_p, C.c_int]
        L.harness_seconds.restype = C.c_double
        L.harness_destroy.argtypes = [C.c_void_p]
        L.harness_set_cut.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.harness_set_cut.restype = None
        L.harness_set_exchange.argtypes = [C.c_void_p, C.c_int, C.c_int, AR, AG]
        L.harness_set_exchange_coop.argtypes = [C.c_void_p, C.c_int, C.c_int, AR, AG, C.c_int]
        L.harness_preproc.argtypes = [C.c_void_p, _fp, _fp, _fp]
        L.harness_run.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, C.POINTER(C.c_ulonglong)]
        L.harness_rotation.argtypes = [C.c_float, C.c_float, C.c_float, _fp, _fp, C.POINTER(C.c_int)]
        L.harness_overlaps.argtypes = [C.c_float] * 4
        L.harness_closest_orthogonal.argtypes = [_fp, _fp]
        L.harness_svd3.argtypes = [_dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(_fp)


class HostDriver:
    """Product driver template over oracle operators."""

    def __init__(self, pct, pcs, lut_res, mse_thr, schedule=0, round_width=1, trim_fraction=0.0, build_lut=True, use_grid=False):
        pct = np.ascontiguousarray(pct, np.float32); pcs = np.ascontiguousarray(pcs, np.float32)
        self._h = C.c_void_p(lib().harness_create_ex(_f(pct), len(pct), _f(pcs), len(pcs), lut_res, mse_thr, schedule, round_width, trim_fraction,
                                                     int(build_lut), int(use_grid)))
        self._cbs = None

    def lut_dims(self):
        d = (C.c_int * 3)()
        lib().harness_lut_dims(self._h, d)
        return tuple(d)

    def lut_set(self, data):
        a = np.ascontiguousarray(data, np.float32).reshape(-1)
        if lib().harness_lut_set(self._h, _f(a), a.size):
            raise ValueError("LUT size mismatch")

    def seconds(self):
        return {k: lib().harness_seconds(self._h, i) for i, k in enumerate(("total", "bnb", "icp"))}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().harness_destroy(self._h)
            self._h = None

    def set_cut(self, driver_passes_thresholds=True, oracle_applies_them=False):
        """Early exit (fgoicp_bounds_submit_cut): whether the tasks' thresholds reach the operator, and whether the oracle operator then
        answers {T, T} for a row at or above its threshold, as the device does."""
        lib().harness_set_cut(self._h, int(bool(driver_passes_thresholds)), int(bool(oracle_applies_them)))

    def set_exchange(self, rank, world, allreduce_min, allgather, coop=False):
        """coop: cooperative rounds (the driver flow a device all-gather switches on; the CPU backend runs every ICP replicated)"""
        self._cbs = (AR(allreduce_min), AG(allgather))
        lib().harness_set_exchange_coop(self._h, rank, world, *self._cbs, int(bool(coop)))

    def preproc(self):
        offs = np.empty(6, np.float32); scale = C.c_float(); b = np.empty(6, np.float32)
        lib().harness_preproc(self._h, _f(offs), C.byref(scale), _f(b))
        return dict(offset_pcs=offs[:3].copy(), offset_pct=offs[3:].copy(), scale=np.float32(scale.value), bounds=b.reshape(3, 2))

    def run(self):
        R = np.empty(9, np.float32); t = np.empty(3, np.float32); ts = np.empty(3, np.float32); sse = C.c_float()
        st = (C.c_ulonglong * 7)()
        rc = lib().harness_run(self._h, _f(R), _f(t), _f(ts), C.byref(sse), st)
        if rc:
            raise RuntimeError(f"driver failed with status {rc}")
        names = ["trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds"]
        return dict(R=R.reshape(3, 3).T.copy(), t=t, t_scaled=ts, best_sse=np.float32(sse.value), stats={n: int(st[i]) for i, n in enumerate(names)})


def point_order(xyz, leaf=64, mode=2, fine=True):
    """csrc/device/morton.hpp: the order a cloud is stored in on the device (1 = Hilbert curve, 2 = k-d cells, 3 = density split)"""
    xyz = np.ascontiguousarray(xyz, np.float32)
    perm = np.empty(len(xyz), np.uint32)
    L = lib()
    L.harness_point_order.argtypes = [_fp, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.harness_point_order.restype = None
    L.harness_point_order(_f(xyz), len(xyz), int(leaf), int(mode), int(bool(fine)), perm.ctypes.data_as(C.POINTER(C.c_uint32)))
    return perm


def slab_d2(n3, a, b, q):
    """csrc/device/slab.hpp: squared slab distance (with its rounding allowance) of the queries q (m, 3) from {p: a <= n.p <= b}"""
    n3 = np.ascontiguousarray(n3, np.float32); q = np.ascontiguousarray(q, np.float32)
    out = np.empty(len(q), np.float32)
    L = lib()
    L.harness_slab_d2.argtypes = [_fp, C.c_float, C.c_float, _fp, C.c_size_t, _fp]
    L.harness_slab_d2.restype = None
    L.harness_slab_d2(_f(n3), float(a), float(b), _f(q), len(q), _f(out))
    return out


def rotation(x, y, z):
    R = np.empty(9, np.float32); r = C.c_float(); ok = C.c_int()
    lib().harness_rotation(x, y, z, _f(R), C.byref(r), C.byref(ok))
    return R.reshape(3, 3).T.copy(), np.float32(r.value), bool(ok.value)


def overlaps(x, y, z, span):
    return bool(lib().harness_overlaps(x, y, z, span))


def closest_orthogonal(ABt9):
    a = np.ascontiguousarray(ABt9, np.float32).reshape(9); out = np.empty(9, np.float32)
    lib().harness_closest_orthogonal(_f(a), _f(out))
    return out


def svd3(A):
    A = np.ascontiguousarray(A, np.float64).reshape(9)
    U = np.empty(9); S = np.empty(3); V = np.empty(9)
    lib().harness_svd3(A.ctypes.data_as(_dp), U.ctypes.data_as(_dp), S.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return U.reshape(3, 3), S, V.reshape(3, 3)
