// tools/store_shape_probe.hip -- HBM write bandwidth of plane-strided f32 stores by block SHAPE (B x 512^2 planes, 32 planes per block):
// does a tile-shaped block (row segments of 128 / 512 bytes, 2 KB apart) write as fast as a block of consecutive pixels?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_shape_probe.hip -o tools/store_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int W = 512, H = 512, NB = 16, PL = 128;
constexpr long long HW = (long long)W * H;

// shape 0: 256 consecutive pixels (1 KB of one row); 1: tile 32 x 8; 2: tile 64 x 4; 3: tile 128 x 2; 4: tile 16 x 16
template <int SH>
__global__ __launch_bounds__(256) void st4(float* out, float v) {
    constexpr int TW = SH == 0 ? 256 : SH == 1 ? 32 : SH == 2 ? 64 : SH == 3 ? 128 : 16, TH = 256 / TW;
    constexpr int tpr = W / TW, tpc = H / TH;
    const int b = blockIdx.x / (tpr * tpc), r = blockIdx.x % (tpr * tpc);
    const int x = (r % tpr) * TW + threadIdx.x % TW, y = (r / tpr) * TH + threadIdx.x / TW;
    float* base = out + ((long long)b * PL + blockIdx.y * 32) * HW + (long long)y * W + x;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) base[c * HW] = v + c;
}
// 16-byte stores: shape 0: 1024 consecutive pixels per block; 1: tile 128 x 8; 2: tile 32 x 32; 3: tile 256 x 4
template <int SH>
__global__ __launch_bounds__(256) void st16(float* out, float v) {
    constexpr int TW = SH == 0 ? 1024 : SH == 1 ? 128 : SH == 2 ? 32 : 256, TH = 1024 / TW;
    if constexpr (SH == 0) {
        const long long p = blockIdx.x * 1024ll + threadIdx.x * 4;
        const int b = (int)(p / HW);
        float* base = out + ((long long)b * PL + blockIdx.y * 32) * HW + p % HW;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) *reinterpret_cast<float4*>(base + c * HW) = make_float4(v + c, v, v, v);
    } else {
        constexpr int tpr = W / TW, tpc = H / TH;
        const int b = blockIdx.x / (tpr * tpc), r = blockIdx.x % (tpr * tpc);
        const int x = (r % tpr) * TW + (threadIdx.x % (TW / 4)) * 4, y = (r / tpr) * TH + threadIdx.x / (TW / 4);
        float* base = out + ((long long)b * PL + blockIdx.y * 32) * HW + (long long)y * W + x;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) *reinterpret_cast<float4*>(base + c * HW) = make_float4(v + c, v, v, v);
    }
}
template <class F>
static void timeit(const char* name, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    printf("%-44s %8.1f us  %5.2f TB/s\n", name, best * 1e3, (double)NB * PL * HW * 4 / best * 1e-9);
}
int main() {
    float* d;
    CK(hipMalloc(&d, (size_t)NB * PL * HW * 4));
    const dim3 g4((unsigned)(NB * HW / 256), PL / 32), g16((unsigned)(NB * HW / 1024), PL / 32);
    timeit("4-byte, 256 consecutive pixels", [&] { hipLaunchKernelGGL(st4<0>, g4, dim3(256), 0, 0, d, 1.f); });
    timeit("4-byte, tile 128 x 2", [&] { hipLaunchKernelGGL(st4<3>, g4, dim3(256), 0, 0, d, 1.f); });
    timeit("4-byte, tile 64 x 4", [&] { hipLaunchKernelGGL(st4<2>, g4, dim3(256), 0, 0, d, 1.f); });
    timeit("4-byte, tile 32 x 8", [&] { hipLaunchKernelGGL(st4<1>, g4, dim3(256), 0, 0, d, 1.f); });
    timeit("4-byte, tile 16 x 16", [&] { hipLaunchKernelGGL(st4<4>, g4, dim3(256), 0, 0, d, 1.f); });
    timeit("16-byte, 1024 consecutive pixels", [&] { hipLaunchKernelGGL(st16<0>, g16, dim3(256), 0, 0, d, 1.f); });
    timeit("16-byte, tile 256 x 4", [&] { hipLaunchKernelGGL(st16<3>, g16, dim3(256), 0, 0, d, 1.f); });
    timeit("16-byte, tile 128 x 8", [&] { hipLaunchKernelGGL(st16<1>, g16, dim3(256), 0, 0, d, 1.f); });
    timeit("16-byte, tile 32 x 32", [&] { hipLaunchKernelGGL(st16<2>, g16, dim3(256), 0, 0, d, 1.f); });
    return 0;
}
