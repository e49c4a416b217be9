// sh16.h -- the split-operand number format shared by every producer / consumer of SH16 tensors.
//
// An f32 value v of a tensor with power-of-two scale s is stored as two f16 numbers
//      hi = f16(v*s),   lo = f16(v*s - hi)
// converted with MODE.FP16_OVFL set (sh16_mode_on): a finite value outside the f16 range becomes +-65504 instead of inf
// (measured on gfx950, tools/fp16_ovfl_test.hip: 70000 -> hi 65504 + lo 4496; inf and NaN are preserved), so no inf can
// reach an MFMA from a finite activation.  hi + lo carries 22 significand bits of v*s as long as lo is a normal f16
// number, i.e. |v*s| >= 2^-3; below that the absolute error is at most 2^-25 (half an f16 subnormal step).  Scales are
// chosen so that this floor is irrelevant:
//   * weights: GEMM rows are scaled by a power of two 2^k with max|row| * 2^k in [2^14, 2^15) (host, at ch_finalize; per row
//     for the plain convs, per 64-row wave tile for the SPADE gamma/beta rows): an element 2^18 times smaller than that
//     maximum still has all 22 bits, the floor is 2^-40 of the maximum;
//   * activations with a bound known before they are computed (label-table sums: a table bound; instance norm: sqrt(HW))
//     use the scale that puts the bound in [2^14, 2^15): they cannot saturate;
//   * data-dependent activations (ACE outputs, style projections) are written with the fixed scale SH16_ACT_SCALE while the
//     producer tracks the exact max |v*s| of the tensor in a device slot.  If that maximum left the window [2^-1, 65504]
//     (f32-class accuracy relative to the tensor's maximum needs max*s >= 2^-1), the producer's second pass -- the same
//     kernel, launched again, returning at once in the normal case -- recomputes the tensor with the scale that puts the
//     recorded maximum in [2^14, 2^15).  Consumers derive the scale in effect from the same slot (sh16_dyn_extra).
// All scalings are by powers of two and are undone exactly in the consumer's f32 epilogue (accumulator * 2^-k / s).
#pragma once
#include <hip/hip_runtime.h>

namespace chk {

constexpr float SH16_MAX = 65504.f;
constexpr float SH16_ACT_SCALE = 8.f;        // first-pass scale of data-dependent activation tensors

// power-of-two scale that brings `bound` (an upper bound of |v|) into [2^14, 2^15); host and device
inline __host__ __device__ float sh16_scale_for_bound(float bound) {
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(bound, &e);                 // bound = m * 2^e, m in [0.5, 1)
    return ldexpf(1.f, 15 - e);
}

// `amax_bits`: float bits of max |v * SH16_ACT_SCALE| recorded by the producer's first pass.  Returns the extra power of
// two the tensor is (re)written with: 1 inside the window, else the factor that moves the maximum into [2^14, 2^15).
inline __host__ __device__ float sh16_dyn_extra(unsigned amax_bits) {
    float a;
    __builtin_memcpy(&a, &amax_bits, 4);
    if (a == 0.f || (a >= 0.5f && a <= SH16_MAX) || !(a < 3.0e38f)) return 1.f;      // all-zero / in window / inf, NaN
    return sh16_scale_for_bound(a);
}

// MODE.FP16_OVFL = 1 for the rest of the wave's life: hwreg(HW_REG_MODE = 1, offset 23, size 1)
__device__ __forceinline__ void sh16_mode_on() { __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1); }

// requires sh16_mode_on() earlier in the kernel
__device__ __forceinline__ void sh16_split(float v, float s, _Float16& h, _Float16& l) {
    const float t = v * s;
    h = (_Float16)t;
    l = (_Float16)(t - (float)h);
}

// bf16 single-term mode (BASELINE.json configs[4]): the hi plane holds the bf16 bit pattern of v*s, the lo plane is unused
__device__ __forceinline__ void sh16_split_any(float v, float s, int bf16, _Float16& h, _Float16& l) {
    if (bf16) {
        const __bf16 b = (__bf16)(v * s);
        h = __builtin_bit_cast(_Float16, b);
        l = (_Float16)0.f;
    } else {
        sh16_split(v, s, h, l);
    }
}

__device__ __forceinline__ float sh16_wave_max(float m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return m;
}

// Record a wave's maximum in a slot.  Same-address atomics serialise in L2 (~13 ns each, measured: 32k waves of an
// elementwise kernel spent 0.38 ms there), so the slot is read first and the atomic only issued by waves that would raise
// it -- a stale read costs an unnecessary atomic, never a missed one (the slot only grows).  Call from one lane per wave.
__device__ __forceinline__ void sh16_slot_max(unsigned* slot, float v) {
    const unsigned b = __float_as_uint(v);                    // v >= 0: float order == unsigned order
    if (b > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, b);
}

// One commit per BLOCK (every thread of the block must call it): elementwise producers launch few, large blocks
// (SH16_EW_BLOCKS x SH16_EW_THREADS, grid-stride) so that a kernel issues a few hundred atomics, not tens of thousands.
constexpr int SH16_EW_THREADS = 1024, SH16_EW_BLOCKS = 512;
__device__ __forceinline__ void sh16_block_slot_max(unsigned* slot, float v) {
    __shared__ float s_wmax[16];
    v = sh16_wave_max(v);
    if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0 && slot) {
        float m = 0.f;
        for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) m = fmaxf(m, s_wmax[w]);
        sh16_slot_max(slot, m);
    }
}
inline int sh16_ew_grid(long long n) {
    const long long b = (n + SH16_EW_THREADS - 1) / SH16_EW_THREADS;
    return (int)(b < SH16_EW_BLOCKS ? (b > 0 ? b : 1) : SH16_EW_BLOCKS);
}

}  // namespace chk
