// v_mfma_f32_32x32x16_f16 issue rate as a function of the number of independent accumulators in flight (dependent
// distance): decides how the conv k-loops may order their MFMAs.  hipcc --offload-arch=gfx950 -O3 tools/mfma_dep.hip -o tools/mfma_dep.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(float* out, int cus) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(256), 0, 0, out, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(cus), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
    printf("independent accumulators %d (1 wave/SIMD): %.2f ms  %.1f TFLOP/s\n", NACC, ms, flops / ms / 1e9);
}
int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    float* out; hipMalloc(&out, (size_t)pr.multiProcessorCount * 256 * 4);
    run<1>(out, pr.multiProcessorCount); run<2>(out, pr.multiProcessorCount); run<4>(out, pr.multiProcessorCount); run<8>(out, pr.multiProcessorCount);
    return 0;
}
