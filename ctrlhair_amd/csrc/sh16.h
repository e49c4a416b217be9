// sh16.h -- the split-operand number format shared by every producer / consumer of SH16 tensors.
//
// An f32 value v of a tensor with power-of-two scale s is stored as two f16 numbers
//      hi = f16(clamp(v*s)),   lo = f16(v*s - hi)            (clamp to the finite f16 range: no inf ever reaches an MFMA)
// hi + lo carries 22 significand bits of v*s as long as lo is a normal f16 number, i.e. |v*s| >= 2^-3; below that the
// absolute error is at most 2^-25 (half an f16 subnormal step).  Scales are chosen so that this floor is irrelevant:
//   * weights: every GEMM row is scaled by its own 2^k with max|row| * 2^k in [2^14, 2^15) (host, at ch_finalize); an
//     element 2^18 times smaller than its row's maximum still has all 22 bits, the floor is 2^-40 of the row maximum;
//   * activations: one power-of-two scale per tensor, a bound where the producer has one (label-table sums, instance norm),
//     otherwise a fixed 2^3 -- values in [2^-4, 8188] keep f32-class relative accuracy, larger ones saturate and are counted
//     (ch_get_stat); the consumer's epilogue multiplies its f32 accumulators by the exact inverse 2^-k / s.
// All scalings are by powers of two, so the represented values are the same as without scaling wherever no floor is hit.
#pragma once
#include <hip/hip_runtime.h>

namespace chk {

constexpr float SH16_MAX = 65504.f;
constexpr float SH16_ACT_SCALE = 8.f;        // fixed scale of data-dependent activation tensors (ACE outputs, style projections)

// power-of-two scale that brings `bound` (an upper bound of |v|) into [2^14, 2^15]; host and device
inline __host__ __device__ float sh16_scale_for_bound(float bound) {
    if (!(bound > 0.f)) return 1.f;
    int e;
    (void)frexpf(bound, &e);                 // bound = m * 2^e, m in [0.5, 1)
    return ldexpf(1.f, 15 - e);
}

__device__ __forceinline__ void sh16_split(float v, float s, _Float16& h, _Float16& l) {
    const float t = __builtin_amdgcn_fmed3f(v * s, -SH16_MAX, SH16_MAX);
    h = (_Float16)t;
    l = (_Float16)(t - (float)h);
}

}  // namespace chk
