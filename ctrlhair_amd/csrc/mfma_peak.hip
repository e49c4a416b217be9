// mfma_peak.hip -- sustained matrix-core issue rate of THIS device, measured live (ch_mfma_peak): an MFMA-only loop with no
// memory traffic, 8 independent 32x32 accumulators per wave, 2 waves per SIMD, random (non-trivial) operand bits so that the
// power draw -- and therefore the clock the part sustains -- resembles a real GEMM main loop.  bench.py reports it next to
// the spec peak: the spec assumes 2.4 GHz, conv launches sit nearer 2.0 GHz.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace chk {

typedef _Float16 pk_half8 __attribute__((ext_vector_type(8)));
typedef float pk_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float pk_rand(unsigned& s) {         // uniform in [-1, 1)
    s = s * 1664525u + 1013904223u;
    return (float)(int)(s >> 8) * (1.f / 8388608.f) - 1.f;
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(float* out, int iters) {
    pk_f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned s = blockIdx.x * 256u + threadIdx.x + 12345u;
    if constexpr (KIND == 0) {
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = pk_rand(s); b[e] = pk_rand(s) * 0.01f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
    } else {
        pk_half8 a[2], b[2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[0][e] = (_Float16)pk_rand(s); a[1][e] = (_Float16)pk_rand(s);
            b[0][e] = (_Float16)(pk_rand(s) * 0.01f); b[1][e] = (_Float16)(pk_rand(s) * 0.01f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 1], b[(i >> 1) & 1], acc[i], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = t;
}

// kind 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_f16.  Runs ~ms_target ms (after a calibration launch); synchronises.
hipError_t mfma_peak(int kind, int ms_target, double* tflops, hipStream_t st) {
    int dev = 0;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    const int blocks = prop.multiProcessorCount * 2;
    float* out = nullptr;
    if ((e = hipMalloc(&out, (size_t)blocks * 256 * sizeof(float))) != hipSuccess) return e;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const double flop_per_iter = (double)blocks * 4 * 8 * 2.0 * 32 * 32 * (kind == 0 ? 2 : 16);
    auto run = [&](int iters, float& ms) {
        (void)hipEventRecord(e0, st);
        if (kind == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(256), 0, st, out, iters);
        else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, st, out, iters);
        (void)hipEventRecord(e1, st);
        hipError_t r = hipEventSynchronize(e1);
        if (r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
        return r;
    };
    float ms = 0.f;
    e = run(2000, ms);                                        // calibration (and warm-up)
    if (e == hipSuccess && ms > 0.f) {
        long long iters = (long long)(2000.0 * ms_target / ms);
        if (iters < 2000) iters = 2000;
        if (iters > 50000000) iters = 50000000;
        e = run((int)iters, ms);
        if (e == hipSuccess && tflops) *tflops = flop_per_iter * (double)iters / (ms * 1e-3) / 1e12;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return e;
}

}  // namespace chk
