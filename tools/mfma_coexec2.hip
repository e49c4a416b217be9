// tools/mfma_coexec2.hip -- follow-up of mfma_coexec.hip: what decides the price of a vector instruction beside v_mfma_f32_16x16x4_f32?
//   ACC: accumulators in the vector half ("v") or in the accumulator half ("a") of the register file (inline-assembly MFMAs);
//   OPS: 1 = v_fma_f32 with one VGPR source (x * s + const), 2 = v_add_f32 with two VGPR sources, 3 = v_fma_f32 with three VGPR sources.
// 8 independent MFMAs + NV vector instructions per iteration, two waves per SIMD (512-thread blocks, one per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec2.hip -o tools/mfma_coexec2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int ACC, int OPS, int WAVES, int RUN = 1>
__global__ __launch_bounds__(WAVES * 64, 1) void loop_kernel(float* out, int iters, float c) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[NV > 0 ? NV : 1], y[NV > 0 ? NV : 1], z[NV > 0 ? NV : 1];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = threadIdx.x * 0.002f - i; z[i] = 0.5f + i; }
    const float a = threadIdx.x * 0.37f + 1.f, b = threadIdx.x * 0.11f - 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8 / RUN; ++g) {
#pragma unroll
            for (int i = g * RUN; i < (g + 1) * RUN; ++i) {
                if (ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int j = g * RUN * NV / 8; j < (g + 1) * RUN * NV / 8; ++j) {
                if (OPS == 1) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[j]) : "s"(c));
                if (OPS == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(y[j]));
                if (OPS == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(z[j]));
                if (OPS == 4) asm volatile("v_fma_f32 %0, 4.0, %1, %0" : "+v"(x[j]) : "v"(y[j]));               // VOP3, inline constant
                if (OPS == 5) asm volatile("v_fmamk_f32 %0, %1, 0xc0a00000, %0" : "+v"(x[j]) : "v"(y[j]));      // VOP2 + 32-bit literal (-5.0)
                if (OPS == 6) asm volatile("v_fmac_f32 %0, 4.0, %1" : "+v"(x[j]) : "v"(y[j]));                  // VOP2 e32, inline constant
                if (OPS == 7) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[j]) : "v"(y[j]));                    // VOP2 e32
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int ACC, int OPS, int WAVES, int RUN = 1>
static void run(float* out, int cus) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 40000;
    hipLaunchKernelGGL((loop_kernel<NV, ACC, OPS, WAVES, RUN>), dim3(cus), dim3(WAVES * 64), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<NV, ACC, OPS, WAVES, RUN>), dim3(cus), dim3(WAVES * 64), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)cus * WAVES * iters * 8 * 2.0 * 16 * 16 * 4;
    const char* on[] = {"", "v_fma (1 VGPR src)", "v_add (2 VGPR src)", "v_fma (3 VGPR src)", "v_fma vop3 inl.const", "v_fmamk literal", "v_fmac e32 inl.const", "v_sub e32"};
    printf("%d waves/SIMD, acc in %s, runs of %d MFMAs, 8 MFMAs + %2d %-20s: %7.2f ms  %6.1f TFLOP/s  (%.0f ns / iteration / SIMD)\n", WAVES / 4, ACC ? "AGPR" : "VGPR", RUN, NV, on[OPS],
           ms, flops / ms / 1e9, ms * 1e6 / iters);
}

int main() {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, (size_t)cus * 512 * 4);
#define ROW(NV, OPS) run<NV, 0, OPS, 8>(out, cus); run<NV, 1, OPS, 8>(out, cus); run<NV, 0, OPS, 4>(out, cus); run<NV, 1, OPS, 4>(out, cus);
    ROW(0, 1)
    ROW(8, 1) ROW(16, 1) ROW(32, 1)
    ROW(16, 2) ROW(32, 2)
    ROW(16, 3) ROW(32, 3)
    ROW(32, 4) ROW(32, 5) ROW(32, 6) ROW(32, 7)
#define RR(NV, R) run<NV, 0, 2, 8, R>(out, cus); run<NV, 0, 2, 4, R>(out, cus);
    RR(8, 1) RR(8, 2) RR(8, 4) RR(8, 8)
    RR(16, 1) RR(16, 2) RR(16, 4) RR(16, 8)
    RR(32, 1) RR(32, 2) RR(32, 4) RR(32, 8)
    return 0;
}
