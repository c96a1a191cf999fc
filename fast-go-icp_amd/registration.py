"""Python mirror of the reference's operator classes for the hot path — same names, argument
meaning and return order as fgoicp/registration.hpp and fgoicp/icp3d.hpp — over the C ABI of
libfgoicp_amd.so.  Every call runs on the MI355X; nothing here computes on the CPU."""
import ctypes as C

import numpy as np

from . import _lib
from .nodes import RotNode, from_glm, pack_tnodes, to_glm


def _fp(a):
    return a.ctypes.data_as(_lib.c_float_p)


def _cloud(pc):
    a = np.ascontiguousarray(pc, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("point cloud must be (n, 3)")
    return a


class StreamPool:
    """Kept for signature compatibility with icp::StreamPool (fgoicp/common.hpp:138-164); the
    context owns its HIP stream, a batch is one fused launch instead of 32 per-stream launches."""

    def __init__(self, size=32):
        self.size = size


def cloud_stats(points):
    """fgoicp_cloud_stats: count, centroid, bounding box, largest centred coordinate, RMS radius of a raw cloud (host side)."""
    p = _cloud(points)
    st = _lib.CloudStats()
    _lib.check(_lib.load().fgoicp_cloud_stats(_fp(p), len(p), C.byref(st)), "fgoicp_cloud_stats")
    return dict(n=int(st.n), centroid=np.array(st.centroid, np.float32), min=np.array(st.min, np.float32), max=np.array(st.max, np.float32),
                max_abs_centred=float(st.max_abs_centred), rms_radius=float(st.rms_radius))


class Registration:
    """icp::Registration (fgoicp/registration.hpp:49-98) + its NearestNeighborLUT member."""

    def __init__(self, pct, pcs, target_bounds, lut_resolution, device=0, flags=0):
        self._lib = _lib.load()
        self.pct = _cloud(pct)
        self.pcs = _cloud(pcs)
        self.nt, self.ns = len(self.pct), len(self.pcs)
        b = np.asarray(target_bounds, dtype=np.float32).reshape(6)  # ((minx,maxx),(miny,maxy),(minz,maxz))
        self._h = C.c_void_p()
        _lib.check(self._lib.fgoicp_ctx_create(_fp(self.pct), self.nt, _fp(self.pcs), self.ns, _fp(b), float(lut_resolution),
                                               int(device), int(flags), C.byref(self._h)), "fgoicp_ctx_create")

    @classmethod
    def _borrow(cls, handle, owner):
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = C.c_void_p(handle)
        self._owner = owner  # keeps the solver alive; this object must not destroy the ctx
        self.ns = self._lib.fgoicp_ctx_ns(self._h)
        self.nt = self._lib.fgoicp_ctx_nt(self._h)
        return self

    def close(self):
        if getattr(self, "_h", None) and not hasattr(self, "_owner"):
            self._lib.fgoicp_ctx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- NearestNeighborLUT ---------------------------------------------------------------
    def lut_dims(self):
        d = (C.c_int * 3)()
        _lib.check(self._lib.fgoicp_lut_dims(self._h, d), "fgoicp_lut_dims")
        return tuple(d)

    def set_coop_split(self, min_points_untrimmed=None, min_points_trimmed=None):
        """fgoicp_ctx_set_coop_split: from how many source points a cooperative refinement splits its scans over the ranks (None = never)"""
        never = (1 << 64) - 1
        _lib.check(self._lib.fgoicp_ctx_set_coop_split(self._h, never if min_points_untrimmed is None else int(min_points_untrimmed),
                                                       never if min_points_trimmed is None else int(min_points_trimmed)), "fgoicp_ctx_set_coop_split")

    def info(self):
        """fgoicp_ctx_get_info: LUT size and layout, source density per LUT face voxel, points per work item (what the context
        derived from the clouds' statistics)."""
        i = _lib.CtxInfo()
        i.struct_size = C.sizeof(_lib.CtxInfo)
        _lib.check(self._lib.fgoicp_ctx_get_info(self._h, C.byref(i)), "fgoicp_ctx_get_info")
        return dict(lut_dims=tuple(i.lut_dims), lut_layout=i.lut_layout, lut_nodes=i.lut_nodes, lut_bytes=i.lut_bytes,
                    source_points_per_face_voxel=i.source_points_per_face_voxel, points_per_item=i.points_per_item,
                    items_per_evaluation=i.items_per_evaluation, max_subcubes_per_window=i.max_subcubes_per_window,
                    source_order=i.source_order, tree_order=i.tree_order, chunks_per_item_with_thresholds=i.chunks_per_item_with_thresholds)

    def lut_read(self):
        dx, dy, dz = self.lut_dims()
        out = np.empty(dx * dy * dz, dtype=np.float32)
        _lib.check(self._lib.fgoicp_lut_read(self._h, _fp(out), out.size), "fgoicp_lut_read")
        return out.reshape(dz, dy, dx)

    def lut_nodes(self, xyz):
        """Single LUT nodes by index (n, 3) int (x, y, z) — for LUTs too large to read back whole."""
        q = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        out = np.empty(len(q), dtype=np.float32)
        _lib.check(self._lib.fgoicp_lut_nodes(self._h, q.ctypes.data_as(_lib.c_int_p), len(q), _fp(out)), "fgoicp_lut_nodes")
        return out

    def lut_search(self, queries):
        q = _cloud(queries)
        out = np.empty(len(q), dtype=np.float32)
        _lib.check(self._lib.fgoicp_lut_search(self._h, _fp(q), len(q), _fp(out)), "fgoicp_lut_search")
        return out

    # -- Registration::compute_sse_error, both overloads ---------------------------------------
    def compute_sse_error(self, *args):
        """(R, t) -> float sse                               registration.hpp:96
        (rnode, tnodes, fix_rot[, stream_pool]) -> (lb, ub)  registration.hpp:97 (lower first)"""
        if isinstance(args[0], RotNode):
            rnode, tnodes, fix_rot = args[0], args[1], args[2]
            return self.compute_bounds(rnode.q.R, rnode.span, tnodes, fix_rot)
        R, t = args
        Rg = to_glm(R)
        tt = np.ascontiguousarray(t, dtype=np.float32).reshape(3)
        out = C.c_float()
        _lib.check(self._lib.fgoicp_sse(self._h, _fp(Rg), _fp(tt), C.byref(out)), "fgoicp_sse")
        return np.float32(out.value)

    def compute_bounds(self, R, rot_span, tnodes, fix_rot):
        Rg = to_glm(R)
        tn = pack_tnodes(tnodes)
        B = len(tn)
        lb = np.empty(B, dtype=np.float32)
        ub = np.empty(B, dtype=np.float32)
        _lib.check(self._lib.fgoicp_bounds_batch(self._h, _fp(Rg), float(rot_span), _fp(tn), B, int(bool(fix_rot)), _fp(lb), _fp(ub)),
                   "fgoicp_bounds_batch")
        return lb, ub

    def compute_bounds_multi(self, Rs, rot_spans, fix_rots, tnode_groups):
        """G rotation nodes in one submission; returns lists of (lb, ub) per group."""
        G = len(Rs)
        Rg = np.concatenate([to_glm(R) for R in Rs]).astype(np.float32) if G else np.zeros(0, np.float32)
        spans = np.asarray(rot_spans, dtype=np.float32)
        fr = np.asarray([int(bool(f)) for f in fix_rots], dtype=np.int32)
        packed = [pack_tnodes(t) for t in tnode_groups]
        offs = np.zeros(G + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(p) for p in packed])
        tn = np.concatenate(packed) if G else np.zeros((0, 4), np.float32)
        lb = np.empty(len(tn), dtype=np.float32)
        ub = np.empty(len(tn), dtype=np.float32)
        _lib.check(self._lib.fgoicp_bounds_multi(self._h, G, _fp(Rg), _fp(spans), fr.ctypes.data_as(_lib.c_int_p),
                                                 offs.ctypes.data_as(_lib.c_int_p), _fp(np.ascontiguousarray(tn)), _fp(lb), _fp(ub)),
                   "fgoicp_bounds_multi")
        return [(lb[offs[g]:offs[g + 1]], ub[offs[g]:offs[g + 1]]) for g in range(G)]

    def compute_bounds_cut(self, Rs, rot_spans, fix_rots, tnode_groups, cut_above, twin=None, slot=0):
        """fgoicp_bounds_submit_cut + fgoicp_bounds_collect: as compute_bounds_multi, but a subcube of group g whose lower bound is
        >= cut_above[g] comes back as lb = ub = cut_above[g] (np.inf: exact).  twin: optional array over all subcubes (-1 = none)."""
        G = len(Rs)
        Rg = np.concatenate([to_glm(R) for R in Rs]).astype(np.float32) if G else np.zeros(0, np.float32)
        spans = np.asarray(rot_spans, dtype=np.float32)
        fr = np.asarray([int(bool(f)) for f in fix_rots], dtype=np.int32)
        packed = [pack_tnodes(t) for t in tnode_groups]
        offs = np.zeros(G + 1, dtype=np.int32)
        offs[1:] = np.cumsum([len(p) for p in packed])
        tn = np.ascontiguousarray(np.concatenate(packed) if G else np.zeros((0, 4), np.float32))
        cut = None if cut_above is None else np.ascontiguousarray(cut_above, dtype=np.float32)
        assert cut is None or len(cut) == G
        tw = None if twin is None else np.ascontiguousarray(twin, dtype=np.int32)
        lb = np.empty(len(tn), dtype=np.float32)
        ub = np.empty(len(tn), dtype=np.float32)
        _lib.check(self._lib.fgoicp_bounds_submit_cut(self._h, int(slot), G, _fp(Rg), _fp(spans), fr.ctypes.data_as(_lib.c_int_p), offs.ctypes.data_as(_lib.c_int_p),
                                                      _fp(tn), None if tw is None else tw.ctypes.data_as(_lib.c_int_p), None if cut is None else _fp(cut)),
                   "fgoicp_bounds_submit_cut")
        _lib.check(self._lib.fgoicp_bounds_collect(self._h, int(slot), _fp(lb), _fp(ub)), "fgoicp_bounds_collect")
        return [(lb[offs[g]:offs[g + 1]], ub[offs[g]:offs[g + 1]]) for g in range(G)]

    def cut_stats(self, reset=False):
        """(work items of the submissions that carried thresholds, work items the early exit did not evaluate) since the last reset"""
        a = C.c_uint64(); b = C.c_uint64()
        _lib.check(self._lib.fgoicp_ctx_cut_stats(self._h, C.byref(a), C.byref(b), int(reset)), "fgoicp_ctx_cut_stats")
        return a.value, b.value

    def procrustes(self, working):
        """One IterativeClosestPoint3D::procrustes() step (icp3d.cu:140-172) on `working` (ns, 3)."""
        w = _cloud(working)
        assert len(w) == self.ns
        R = np.empty(9, np.float32); t = np.empty(3, np.float32); cen = np.empty(6, np.float32); ABt = np.empty(9, np.float32)
        idx = np.empty(self.ns, np.int32)
        _lib.check(self._lib.fgoicp_procrustes(self._h, _fp(w), _fp(R), _fp(t), _fp(cen), _fp(ABt), idx.ctypes.data_as(_lib.c_int_p)),
                   "fgoicp_procrustes")
        return from_glm(R), t, cen, ABt, idx

    def set_inliers(self, k):
        """EXTENSION (trimmed Go-ICP): every sum over source points runs over the k smallest terms; 0 = off."""
        _lib.check(self._lib.fgoicp_ctx_set_inliers(self._h, int(k)), "fgoicp_ctx_set_inliers")

    def point_distances(self, R, rot_span, tnode, fix_rot):
        """Trimmed mode, diagnostic: e_i = max(distance_i, 0) of registration.cu:48-52 for one subcube, caller order."""
        tn = pack_tnodes(np.asarray(tnode, np.float32).reshape(1, 4))
        out = np.empty(self.ns, dtype=np.float32)
        _lib.check(self._lib.fgoicp_bounds_point_distances(self._h, _fp(to_glm(R)), float(rot_span), _fp(tn), int(bool(fix_rot)), _fp(out)),
                   "fgoicp_bounds_point_distances")
        return out

    def test_sort_fault(self, nth_tick):
        """TEST HOOK: spoil the nth sorted tick from now (0 = off) so that the on-device permutation check has something to find."""
        _lib.check(self._lib.fgoicp_ctx_test_sort_fault(self._h, int(nth_tick)), "fgoicp_ctx_test_sort_fault")

    def sort_fallbacks(self):
        """(sorted ticks, ticks repeated after a failed permutation check)"""
        a = C.c_uint64(); b = C.c_uint64()
        _lib.check(self._lib.fgoicp_ctx_sort_fallbacks(self._h, C.byref(a), C.byref(b)), "fgoicp_ctx_sort_fallbacks")
        return a.value, b.value

    def trim_stats(self, reset=False):
        """trimmed bounds: (rows selected, rows done again in two passes after the sampled bracket failed its check, bracket members)"""
        out = (C.c_uint64 * 3)()
        _lib.check(self._lib.fgoicp_ctx_trim_stats(self._h, out, int(reset)), "fgoicp_ctx_trim_stats")
        return int(out[0]), int(out[1]), int(out[2])

    def set_profile(self, enabled):
        _lib.check(self._lib.fgoicp_ctx_set_profile(self._h, int(bool(enabled))), "fgoicp_ctx_set_profile")

    def profile(self, reset=False):
        ms = C.c_double(); launches = C.c_uint64(); sub = C.c_uint64(); ev = C.c_uint64(); sel = C.c_double()
        _lib.check(self._lib.fgoicp_ctx_profile_evaluations(self._h, C.byref(ev)), "fgoicp_ctx_profile_evaluations")
        _lib.check(self._lib.fgoicp_ctx_profile_select_ms(self._h, C.byref(sel)), "fgoicp_ctx_profile_select_ms")
        _lib.check(self._lib.fgoicp_ctx_profile(self._h, C.byref(ms), C.byref(launches), C.byref(sub), int(reset)), "fgoicp_ctx_profile")
        return {"kernel_ms": ms.value, "launches": launches.value, "subcubes": sub.value, "evaluations": ev.value, "select_ms": sel.value}


def icp_batch(reg, Rs, ts, max_iter=100, convergence_threshold=0.005):
    """n IterativeClosestPoint3D runs at once (fgoicp_icp_batch): -> (sse (n,), R (n,3,3), t (n,3), iterations (n,))"""
    n = len(Rs)
    R0 = np.concatenate([to_glm(R) for R in Rs]).astype(np.float32) if n else np.zeros(0, np.float32)
    t0 = np.ascontiguousarray(np.asarray(ts, np.float32).reshape(-1))
    sse = np.empty(n, np.float32); Ro = np.empty(9 * n, np.float32); to = np.empty(3 * n, np.float32); it = np.empty(n, np.int32)
    _lib.check(reg._lib.fgoicp_icp_batch(reg._h, n, _fp(R0), _fp(t0), int(max_iter), float(convergence_threshold), _fp(sse), _fp(Ro), _fp(to),
                                         it.ctypes.data_as(_lib.c_int_p)), "fgoicp_icp_batch")
    return sse, np.stack([from_glm(Ro[9 * i:9 * i + 9]) for i in range(n)]) if n else np.zeros((0, 3, 3), np.float32), to.reshape(n, 3), it


class IterativeClosestPoint3D:
    """icp::IterativeClosestPoint3D (fgoicp/icp3d.hpp:9-41): ctor arguments as the reference's
    (the clouds live in `reg`), run() -> (sse, R, t)."""

    def __init__(self, reg, pct=None, pcs=None, max_iter=100, convergence_threshold=0.05, R=None, t=None):
        self.reg = reg
        self.max_iter = int(max_iter)
        self.thr = float(convergence_threshold)
        self.R = np.eye(3, dtype=np.float32) if R is None else np.asarray(R, dtype=np.float32)
        self.t = np.zeros(3, dtype=np.float32) if t is None else np.asarray(t, dtype=np.float32)
        self.iterations = 0

    def run(self):
        lib = self.reg._lib
        sse = C.c_float(); iters = C.c_int()
        R = np.empty(9, np.float32); t = np.empty(3, np.float32)
        _lib.check(lib.fgoicp_icp(self.reg._h, _fp(to_glm(self.R)), _fp(np.ascontiguousarray(self.t, np.float32)), self.max_iter, self.thr,
                                  C.byref(sse), _fp(R), _fp(t), C.byref(iters)), "fgoicp_icp")
        self.iterations = iters.value
        return np.float32(sse.value), from_glm(R), t
