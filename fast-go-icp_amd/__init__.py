"""fgoicp_amd — MI355X-native (gfx950) Go-ICP hot path behind the reference's operator interface.

    Registration / IterativeClosestPoint3D / FastGoICP   mirror icp::* of solemnwind/fast-go-icp
    libfgoicp_amd.so                                    C ABI (include/fgoicp_amd.h), hand-written HIP

There is no CPU implementation in this package: every operator raises if the HIP library is
missing or no MI355X is visible."""
from . import _lib, build, nodes, synth  # noqa: F401
from ._lib import (FLAG_BRUTE_FORCE_NN, FLAG_CURVE_ORDER, FLAG_NO_MORTON, FLAG_NO_WEIGHT_QUANT, FLAG_PROFILE, SCHEDULE_ROUND, SCHEDULE_SERIAL, TRANSPORT_IN_PROCESS,  # noqa: F401
                   TRANSPORT_RCCL, FgoicpError, dev_knobs)
from .registration import IterativeClosestPoint3D, Registration, StreamPool, cloud_stats, icp_batch  # noqa: F401
from .nodes import Rotation, RotNode, TransNode, from_glm, to_glm  # noqa: F401

try:  # the driver mirror needs nothing beyond the C ABI, but keep import errors local to it
    from .fgoicp import FastGoICP  # noqa: F401
    from .multi import MultiGoICP, RcclExchange, rccl_unique_id  # noqa: F401
except ImportError:  # pragma: no cover
    pass
