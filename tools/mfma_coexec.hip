// tools/mfma_coexec.hip -- do plain VALU instructions overlap with MFMAs on gfx950?  An MFMA-only loop (8 x v_mfma_f32_16x16x4_f32 per
// iteration, 256 matrix-pipe cycles) with NV independent v_fma_f32 (or s_add_u32, or ds_read_b128) added per iteration, one and two waves per SIMD: if the adds ran
// under the MFMAs the time would stay flat until the issue port saturates.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_coexec.hip -o tools/mfma_coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: v_fma_f32; 1: s_add_u32 (scalar ALU); 2: ds_read_b128 (conflict-free, results unused until the end)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, float c) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 2];
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i;
    __syncthreads();
    unsigned sacc = 0;
    f32x4 lsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float x[NV > 0 ? NV : 1];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) x[i] = threadIdx.x * 0.001f + i;
    const float a = threadIdx.x * 0.37f + 1.f, b = threadIdx.x * 0.11f - 2.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = i * NV / 8; j < (i + 1) * NV / 8; ++j) {
                if (KIND == 0) x[j] = __builtin_fmaf(x[j], c, 0.5f);
                if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
                if (KIND == 2) {
                    f32x4 r;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"((unsigned)(threadIdx.x * 16 + (j & 1) * 4096)));
                    if (j == NV - 1 && it == iters - 1) { asm volatile("s_waitcnt lgkmcnt(0)"); lsum = r; }
                }
            }
            if (KIND == 2 && i == 7 && (it & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += x[i];
    s += (float)sacc + lsum.x + lsum.w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int KIND>
static void run(float* out, int cus) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int iters = 40000;
        hipLaunchKernelGGL((loop_kernel<NV, KIND>), dim3(cus * bpc), dim3(256), 0, 0, out, iters, 1.0001f);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<NV, KIND>), dim3(cus * bpc), dim3(256), 0, 0, out, iters, 1.0001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * bpc * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("8 MFMAs + %2d %s per iteration, %d wave(s) per SIMD: %7.2f ms  %6.1f TFLOP/s of MFMA  (%.1f ns per iteration per wave)\n", NV,
               KIND == 0 ? "v_fma_f32" : (KIND == 1 ? "s_add_u32" : "ds_read_b128"), bpc, ms,
               flops / ms / 1e9, ms * 1e6 / iters / bpc);
    }
}

int main() {
    hipDeviceProp_t pr;
    (void)hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    float* out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    run<0, 0>(out, cus);
    run<8, 0>(out, cus);
    run<16, 0>(out, cus);
    run<32, 0>(out, cus);
    run<64, 0>(out, cus);
    run<16, 1>(out, cus);
    run<32, 1>(out, cus);
    run<64, 1>(out, cus);
    run<8, 2>(out, cus);
    run<16, 2>(out, cus);
    run<32, 2>(out, cus);
    return 0;
}
