// tools/mfma_f32_shapes.hip -- sustained issue rate of the two exact-f32 MFMA shapes (no memory traffic), random operand bits:
// v_mfma_f32_32x32x2_f32 (16 passes) against v_mfma_f32_16x16x4_f32 (8 passes), one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_shapes.hip -o tools/mfma_f32_shapes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float rnd(unsigned s) { s = s * 1664525u + 1013904223u; s ^= s >> 13; s *= 2654435761u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; }

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    float a[4], b[4];
    for (int e = 0; e < 4; ++e) { a[e] = rnd(threadIdx.x * 8 + e + blockIdx.x * 4096); b[e] = rnd(threadIdx.x * 8 + e + 4); }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 3], b[i & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int bpc = 1; bpc <= 2; ++bpc) {
            const int iters = shape == 0 ? 60000 : 15000;      // the same FLOPs per launch: 8 x 4096 / 32 x 2048 per iteration... x4
            for (int rep = 0; rep < 2; ++rep) {
                if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(cus * bpc), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(cus * bpc), dim3(256), 0, 0, out, iters);
            }
            (void)hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(cus * bpc), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(mfma_loop<1>, dim3(cus * bpc), dim3(256), 0, 0, out, iters);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double per_it = shape == 0 ? 8 * 2.0 * 32 * 32 * 2 : 32 * 2.0 * 16 * 16 * 4;
            const double flops = (double)cus * bpc * 4 * iters * per_it;
            printf("%s  waves/SIMD %d  %.2f ms  %.1f TFLOP/s\n", shape == 0 ? "v_mfma_f32_32x32x2_f32 " : "v_mfma_f32_16x16x4_f32", bpc, ms, flops / ms / 1e9);
        }
    return 0;
}
