"""ctypes binding of libfgoicp_amd.so (include/fgoicp_amd.h).  Loading fails loudly when the
library has not been built; there is no Python or CPU fallback for any operator."""
import ctypes as C
import os

from . import build as _build

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class CtxInfo(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("lut_dims", C.c_int * 3), ("lut_layout", C.c_int), ("lut_nodes", C.c_uint64), ("lut_bytes", C.c_uint64),
                ("source_points_per_face_voxel", C.c_double), ("points_per_item", C.c_int), ("items_per_evaluation", C.c_int),
                ("max_subcubes_per_window", C.c_int), ("source_order", C.c_int), ("tree_order", C.c_int), ("chunks_per_item_with_thresholds", C.c_int)]


class CloudStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("centroid", C.c_float * 3), ("min", C.c_float * 3), ("max", C.c_float * 3), ("max_abs_centred", C.c_float),
                ("rms_radius", C.c_float)]


class Exchange(C.Structure):
    ALLREDUCE_MIN = C.CFUNCTYPE(C.c_int, c_float_p, C.c_size_t, C.c_void_p)
    ALLGATHER = C.CFUNCTYPE(C.c_int, c_float_p, c_float_p, C.c_size_t, C.c_void_p)
    ALLGATHER_DEVICE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)
    _fields_ = [("struct_size", C.c_size_t), ("rank", C.c_int), ("world_size", C.c_int), ("allreduce_min", ALLREDUCE_MIN),
                ("allgather", ALLGATHER), ("user", C.c_void_p), ("allgather_device", ALLGATHER_DEVICE)]  # the last one optional (NULL: no cooperative ICP)

    def __init__(self, rank=0, world_size=1, allreduce_min=None, allgather=None, user=None, allgather_device=None):
        super().__init__()
        self.struct_size = C.sizeof(Exchange)  # ABI 2: the library reads no byte beyond it
        self.rank, self.world_size, self.user = rank, world_size, user
        if allreduce_min is not None:
            self.allreduce_min = allreduce_min
        if allgather is not None:
            self.allgather = allgather
        if allgather_device is not None:
            self.allgather_device = allgather_device


class SolverOpts(C.Structure):
    _fields_ = [("schedule", C.c_int), ("round_width", C.c_int), ("ctx_flags", C.c_uint), ("device", C.c_int), ("trim_fraction", C.c_float)]


class RunStats(C.Structure):
    _fields_ = [("trans_cubes", C.c_uint64), ("bounds_calls", C.c_uint64), ("rot_cubes", C.c_uint64),
                ("icp_runs", C.c_uint64), ("icp_iters", C.c_uint64), ("inner_bnb", C.c_uint64),
                ("rounds", C.c_uint64), ("seconds_total", C.c_double), ("seconds_bnb", C.c_double),
                ("seconds_icp", C.c_double), ("initial_icp_sse", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


FLAG_NO_WEIGHT_QUANT = 1 << 0
FLAG_NO_MORTON = 1 << 1
FLAG_PROFILE = 1 << 2
FLAG_BRUTE_FORCE_NN = 1 << 3
FLAG_CURVE_ORDER = 1 << 4
SCHEDULE_SERIAL = 0
SCHEDULE_ROUND = 1

_SIGS = {
    "fgoicp_last_error": (C.c_char_p, []),
    "fgoicp_version": (C.c_char_p, []),
    "fgoicp_dev_knobs": (C.c_int, []),
    "fgoicp_abi_version": (C.c_int, []),
    "fgoicp_ctx_create": (C.c_int, [c_float_p, C.c_size_t, c_float_p, C.c_size_t, c_float_p, C.c_float, C.c_int, C.c_uint,
                                    C.POINTER(C.c_void_p)]),
    "fgoicp_ctx_destroy": (None, [C.c_void_p]),
    "fgoicp_lut_dims": (C.c_int, [C.c_void_p, c_int_p]),
    "fgoicp_ctx_get_info": (C.c_int, [C.c_void_p, C.POINTER(CtxInfo)]),
    "fgoicp_cloud_stats": (C.c_int, [c_float_p, C.c_size_t, C.POINTER(CloudStats)]),
    "fgoicp_lut_read": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t]),
    "fgoicp_lut_search": (C.c_int, [C.c_void_p, c_float_p, C.c_size_t, c_float_p]),
    "fgoicp_lut_nodes": (C.c_int, [C.c_void_p, c_int_p, C.c_size_t, c_float_p]),
    "fgoicp_bounds_point_distances": (C.c_int, [C.c_void_p, c_float_p, C.c_float, c_float_p, C.c_int, c_float_p]),
    "fgoicp_ctx_sort_fallbacks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fgoicp_ctx_test_sort_fault": (C.c_int, [C.c_void_p, C.c_int]),
    "fgoicp_bounds_batch": (C.c_int, [C.c_void_p, c_float_p, C.c_float, c_float_p, C.c_int, C.c_int, c_float_p, c_float_p]),
    "fgoicp_bounds_multi": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_int_p, c_int_p, c_float_p, c_float_p,
                                      c_float_p]),
    "fgoicp_bounds_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p, c_int_p, c_float_p]),
    "fgoicp_bounds_submit_twins": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p, c_int_p, c_float_p, c_int_p]),
    "fgoicp_bounds_submit_cut": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p, c_int_p, c_float_p, c_int_p, c_float_p]),
    "fgoicp_ctx_cut_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "fgoicp_solver_set_early_exit": (C.c_int, [C.c_void_p, C.c_int]),
    "fgoicp_bounds_collect": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p]),
    "fgoicp_sse": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "fgoicp_icp": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_size_t, C.c_float, c_float_p, c_float_p, c_float_p, c_int_p]),
    "fgoicp_icp_batch": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, C.c_size_t, C.c_float, c_float_p, c_float_p, c_float_p, c_int_p]),
    "fgoicp_procrustes": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_int_p]),
    "fgoicp_ctx_set_inliers": (C.c_int, [C.c_void_p, C.c_size_t]),
    "fgoicp_ctx_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "fgoicp_ctx_profile_evaluations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "fgoicp_ctx_profile_select_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "fgoicp_ctx_trim_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "fgoicp_ctx_set_coop_split": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "fgoicp_ctx_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "fgoicp_ctx_ns": (C.c_size_t, [C.c_void_p]),
    "fgoicp_ctx_nt": (C.c_size_t, [C.c_void_p]),
    "fgoicp_solver_create": (C.c_int, [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_float, C.c_float,
                                       C.POINTER(SolverOpts), C.POINTER(C.c_void_p)]),
    "fgoicp_solver_destroy": (None, [C.c_void_p]),
    "fgoicp_solver_set_exchange": (C.c_int, [C.c_void_p, C.POINTER(Exchange)]),
    "fgoicp_solver_set_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgoicp_solver_run": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "fgoicp_solver_best_error": (C.c_int, [C.c_void_p, c_float_p]),
    "fgoicp_solver_best_transform": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "fgoicp_solver_last_transform": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "fgoicp_solver_stats": (C.c_int, [C.c_void_p, C.POINTER(RunStats)]),
    "fgoicp_solver_preproc": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "fgoicp_solver_ctx": (C.c_void_p, [C.c_void_p]),
    "fgoicp_rccl_library": (C.c_char_p, []),
    "fgoicp_rccl_unique_id": (C.c_int, [C.POINTER(C.c_ubyte)]),
    "fgoicp_rccl_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_void_p)]),
    "fgoicp_rccl_create_ex": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "fgoicp_rccl_test_inprogress": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "fgoicp_rccl_exchange": (C.c_int, [C.c_void_p, C.POINTER(Exchange)]),
    "fgoicp_rccl_calls": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "fgoicp_rccl_abort": (C.c_int, [C.c_void_p]),
    "fgoicp_rccl_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "fgoicp_rccl_destroy": (None, [C.c_void_p]),
    "fgoicp_multi_create": (C.c_int, [c_float_p, C.c_size_t, c_float_p, C.c_size_t, C.c_float, C.c_float, C.POINTER(SolverOpts), c_int_p, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "fgoicp_multi_destroy": (None, [C.c_void_p]),
    "fgoicp_multi_run": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "fgoicp_multi_icp": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_size_t, C.c_float, c_float_p, c_float_p, c_float_p, c_int_p]),
    "fgoicp_multi_world": (C.c_int, [C.c_void_p]),
    "fgoicp_multi_solver": (C.c_void_p, [C.c_void_p, C.c_int]),
    "fgoicp_multi_seconds": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "fgoicp_multi_set_record": (C.c_int, [C.c_void_p, C.c_int]),
    "fgoicp_multi_recorded": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "fgoicp_multi_test_fault": (C.c_int, [C.c_void_p, C.c_int, C.c_long]),
    "fgoicp_multi_replay_rank": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
}
TRANSPORT_RCCL = 0
TRANSPORT_IN_PROCESS = 1

_lib = None


def lib_path():
    return _build.LIB_PATH


def load():
    """Returns the loaded library; raises if it is missing (build it with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built (python __graft_entry__.py build). "
            "fgoicp_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here = the ABI header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dev_knobs():
    """True if the loaded library is the development build (reads the FGOICP_* A/B knobs; csrc/host/knobs.hpp)."""
    return bool(load().fgoicp_dev_knobs())


def exported_symbols():
    return sorted(_SIGS)


class FgoicpError(RuntimeError):
    def __init__(self, status, where):
        lib = load()
        msg = lib.fgoicp_last_error().decode(errors="replace")
        super().__init__(f"{where} failed with status {status}: {msg}")
        self.status = status


def check(status, where):
    if status != 0:
        raise FgoicpError(status, where)
