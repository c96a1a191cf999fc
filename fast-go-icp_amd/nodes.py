"""Host-side value types of the reference (fgoicp/common.hpp:30-128), in numpy.

Matrices are numpy (3, 3) arrays in the mathematical convention M[row, col]; `to_glm` / `from_glm`
convert to the 9-float column-major order of glm::mat3 that the C ABI uses."""
from dataclasses import dataclass, field

import numpy as np

M_INF = np.float32(1e10)
M_SQRT3 = np.float32(1.732050807568877)
M_PI = np.float32(3.141592653589793)


def to_glm(R):
    return np.ascontiguousarray(np.asarray(R, dtype=np.float32).T).reshape(9)


def from_glm(flat):
    return np.asarray(flat, dtype=np.float32).reshape(3, 3).T.copy()


def _f(x):
    return np.float32(x)


class Rotation:
    """fgoicp/common.hpp:30-69 — (x, y, z) is the vector part of a unit quaternion, w >= 0.
    Outside the unit ball R stays identity and r keeps the SQUARED norm (reference quirk)."""

    def __init__(self, x=0.0, y=0.0, z=0.0):
        x, y, z = _f(x), _f(y), _f(z)
        self.x, self.y, self.z = x, y, z
        r = _f(_f(_f(x * x) + _f(y * y)) + _f(z * z))
        self.R = np.eye(3, dtype=np.float32)
        self.r = r
        if r > _f(1.0):
            return
        ww = _f(_f(1.0) - r)
        w = np.sqrt(ww, dtype=np.float32)
        wx, xx = _f(w * x), _f(x * x)
        wy, xy, yy = _f(w * y), _f(x * y), _f(y * y)
        wz, xz, yz, zz = _f(w * z), _f(x * z), _f(y * z), _f(z * z)
        two = _f(2.0)
        # glm::mat3(9 scalars) fills columns: these are the COLUMNS of R
        c0 = [_f(_f(_f(ww + xx) - yy) - zz), _f(two * _f(xy - wz)), _f(two * _f(xz + wy))]
        c1 = [_f(two * _f(xy + wz)), _f(_f(_f(ww - xx) + yy) - zz), _f(two * _f(yz - wx))]
        c2 = [_f(two * _f(xz - wy)), _f(two * _f(yz + wx)), _f(_f(_f(ww - xx) - yy) + zz)]
        self.R = np.array([c0, c1, c2], dtype=np.float32).T.copy()
        self.r = np.sqrt(r, dtype=np.float32)

    def in_SO3(self):
        return bool(self.r <= _f(1.0))


@dataclass
class RotNode:
    """fgoicp/common.hpp:75-104"""
    x: float
    y: float
    z: float
    span: float
    lb: float = 0.0
    ub: float = 0.0
    q: Rotation = field(init=False)

    def __post_init__(self):
        self.q = Rotation(self.x, self.y, self.z)
        self.span = _f(self.span)

    def overlaps_SO3(self):
        q, s = self.q, self.span
        a = _f(_f(abs(q.x) + abs(q.y)) + abs(q.z))
        v = _f(_f(q.r - _f(_f(_f(2.0) * s) * a)) + _f(_f(_f(3.0) * s) * s))
        return bool(v <= _f(1.0))

    def __lt__(self, other):  # std::priority_queue order, common.hpp:85-92
        if self.lb == other.lb:
            return self.span < other.span
        return self.lb > other.lb


@dataclass
class TransNode:
    """fgoicp/common.hpp:110-128"""
    x: float
    y: float
    z: float
    span: float
    lb: float = 0.0
    ub: float = 0.0

    @property
    def t(self):
        return np.array([self.x, self.y, self.z], dtype=np.float32)

    def __lt__(self, other):
        if self.lb == other.lb:
            return self.span < other.span
        return self.lb > other.lb


def pack_tnodes(tnodes):
    """B x {t.x, t.y, t.z, span} float32 rows for the C ABI."""
    if isinstance(tnodes, np.ndarray):
        a = np.ascontiguousarray(tnodes, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        return a
    return np.array([[t.x, t.y, t.z, t.span] for t in tnodes], dtype=np.float32).reshape(-1, 4)
