// mfma_peak.hip -- sustained v_mfma_f32_32x32x16_f16 issue rate of the device (no memory traffic): calibrates how far
// the conv main loops are from what the matrix cores deliver under sustained load (clocks drop under MFMA power).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* out;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000 * (rep + 1);       // longer runs expose the sustained (power-limited) clock
            hipLaunchKernelGGL(mfma_loop<8>, dim3(cus * bpc), dim3(256), 0, 0, out, 1000);
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<8>, dim3(cus * bpc), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)cus * bpc * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
            printf("CUs %d  waves/SIMD %d  iters %d  %.2f ms  %.1f TFLOP/s f16 dense\n", cus, bpc, iters, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
