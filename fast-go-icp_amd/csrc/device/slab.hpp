// The leaf-slab test of the exact nearest-neighbour scans (kernels.hip box_walk; bvh.hip builds the slabs), as a function the host can
// compile too: tests/test_host_logic.py checks its rounding allowance against exact arithmetic.
#pragma once
#include <cmath>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FGOICP_HD __host__ __device__
#else
#define FGOICP_HD
#endif

namespace fgoicp {

// Squared distance of q from the slab {p : a <= n.p <= b} along n — a LOWER bound of the squared distance of q to every point of the
// slab, because |n| <= 1 (bvh.hip shrinks the normal by 1e-6) — minus an allowance for the rounding of n.q, so that it stays a lower
// bound in fp32.  n.q = fma(nz, qz, fma(ny, qy, nx * qx)) carries an absolute error of at most 3 * 2^-24 * (|nx qx| + |ny qy| + |nz qz|)
// (three roundings, each relative to a partial sum that the sum of magnitudes bounds), the subtraction another 2^-24 relative to
// max(|n.q|, |a|, |b|).  The allowance is therefore taken relative to the magnitudes that were actually rounded — round 3 took it
// relative to max(|n.q|, |a|, |b|), which collapses when the three products cancel (a plane through or near the origin, a cloud with a
// large common offset): the slab distance was then over-estimated by up to 1e-7 |q| and a leaf holding the nearest point could be
// rejected.  4e-7 > (3 + 1) * 2^-24 = 2.4e-7.
FGOICP_HD inline float slab_d2(float nx, float ny, float nz, float a, float b, float qx, float qy, float qz) {
    const float nq = fmaf(nz, qz, fmaf(ny, qy, nx * qx));
    const float mag = fmaf(fabsf(nz), fabsf(qz), fmaf(fabsf(ny), fabsf(qy), fabsf(nx) * fabsf(qx)));  // >= |n.q|: |nx qx| + |ny qy| + |nz qz|
    float s = fmaxf(nq - b, a - nq);
    s -= 4e-7f * (mag + fmaxf(fabsf(a), fabsf(b)));
    return s > 0.0f ? s * s : 0.0f;
}

}  // namespace fgoicp
