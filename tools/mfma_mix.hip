// tools/mfma_mix.hip -- how much non-MFMA work fits in the shadow of v_mfma_f32_16x16x4_f32 on gfx950?
// 512-thread blocks (2 waves / SIMD), 32 independent accumulators per wave, NV VALU ops / NS SALU ops / NL ds_read_b128 per 32 MFMAs,
// optional s_barrier per iteration.  Prints the f32 MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NV, int NL, bool BAR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void mix_kernel(float* out, int iters, float seed) {
    __shared__ f32x4 lds[2048];
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 3.f - threadIdx.x;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed * i + threadIdx.x;
    lds[threadIdx.x] = (f32x4){a, b, a, b};
    __syncthreads();
    f32x4 l[NL > 0 ? NL : 1];
#pragma unroll
    for (int i = 0; i < (NL > 0 ? NL : 1); ++i) l[i] = (f32x4){a, b, a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < NL / 4; ++i) l[g * (NL / 4) + i] = lds[(threadIdx.x + 64 * (g * (NL / 4) + i) + it) & 2047];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[g * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + (NL ? l[(g * 8 + i) % NL].x : 0.f), b, acc[g * 8 + i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NV / 4; ++i) v[i % 8] = v[i % 8] * 1.0001f + v[(i + 3) % 8];
        }
        if (BAR) __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, bool BAR, int WAVES>
void run(const char* name) {
    float* out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix_kernel<NV, NL, BAR, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, 100, 1.f);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mix_kernel<NV, NL, BAR, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, iters, 1.f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 256.0 * WAVES * iters * 32 * 2048.0;
    printf("%-44s %7.3f ms  %6.1f TF/s\n", name, ms, fl / ms * 1e-9);
    CK(hipFree(out));
}

int main() {
    run<0, 0, false, 8>("8 waves, MFMA only");
    run<0, 0, false, 4>("4 waves, MFMA only");
    run<16, 0, false, 8>("8 waves, 16 VALU / 32 MFMA");
    run<32, 0, false, 8>("8 waves, 32 VALU / 32 MFMA");
    run<64, 0, false, 8>("8 waves, 64 VALU / 32 MFMA");
    run<128, 0, false, 8>("8 waves, 128 VALU / 32 MFMA");
    run<32, 0, false, 4>("4 waves, 32 VALU / 32 MFMA");
    run<64, 0, false, 4>("4 waves, 64 VALU / 32 MFMA");
    run<0, 8, false, 8>("8 waves, 8 ds_read_b128 / 32 MFMA");
    run<0, 16, false, 8>("8 waves, 16 ds_read_b128 / 32 MFMA");
    run<32, 8, false, 8>("8 waves, 32 VALU + 8 LDS");
    run<32, 8, true, 8>("8 waves, 32 VALU + 8 LDS + barrier");
    run<0, 0, true, 8>("8 waves, MFMA + barrier");
    return 0;
}
