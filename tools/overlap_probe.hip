// tools/overlap_probe.hip -- does a store-bound kernel on a few CUs run under an MFMA-bound persistent conv on the rest?
//   The product's wino_plain_kernel (static task split, grid = G) on stream A, a narrow persistent writer of
// Gs blocks x 1024 threads (plane-strided 16-byte stores, the label-table kernel's pattern) on stream B.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/overlap_probe.hip -o tools/overlap_probe.bin
#include "../ctrlhair_amd/csrc/conv_inst_wino.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace chk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// planes [P][HW] f32: block-task = (plane group of 32, 2048 consecutive pixels); every thread writes float4 to 8 planes... the
// label-table pattern: a wave writes 1 KB runs of one plane, 32 planes 1 MB apart per task
__global__ __launch_bounds__(1024) void store_kernel(float* out, long long hw, int planes, long long ntasks, float v) {
    const int tid = threadIdx.x;
    for (long long t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const long long per = hw / 4096;                 // pixel runs of 4096 per plane group
        const long long pg = t / per, run = t - pg * per;
        float* base = out + pg * 32 * hw + run * 4096 + (long long)tid * 4;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const f32x4 val = {v + c, v, v, v};
            *reinterpret_cast<f32x4*>(base + c * hw) = val;
        }
    }
    (void)planes;
}

int main(int argc, char** argv) {
    const int B = 16, Cin = 128, Cout = 128, H = 256, W = 256;       // up_2 conv_1
    const size_t nin = (size_t)B * Cin * H * W, nout = (size_t)B * Cout * H * W;
    std::vector<float> hin(nin), hw((size_t)Cout * Cin * 9);
    unsigned s = 1234;
    auto fr = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& v : hin) v = fr();
    for (auto& v : hw) v = fr() / 34.f;
    const float* wp = hw.data();
    std::vector<float> pk = pack_wino_A(Cout, Cin, [&](int row, int ci, int t) { return wp[((size_t)row * Cin + ci) * 9 + t]; });
    float *d_in, *d_pk, *d_out, *d_zero, *d_st;
    CK(hipMalloc(&d_in, nin * 4 + 256)); CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_out, nout * 4)); CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    CK(hipMemcpy(d_in, hin.data(), nin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    const long long shw = 16ll * 512 * 512;            // B x 512^2 pixels per plane
    const int planes = 128;
    CK(hipMalloc(&d_st, (size_t)planes * shw * 4));     // 2.1 GB
    const long long stasks = (planes / 32) * (shw / 4096);
    WinoParams p{};
    p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.act = ACT_NONE; p.zero = d_zero;
    wino_fill_launch(p);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_plain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino::LDS_BYTES));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi));
    hipEvent_t e0, ea, eb, ea0, eb0;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreate(&ea0)); CK(hipEventCreate(&eb0));
    const int NC = 4;                                    // conv launches per measurement (~2 ms each)
    auto conv = [&](int G) { for (int i = 0; i < NC; ++i) hipLaunchKernelGGL(wino_plain_kernel<0>, dim3(G), dim3(512), wino::LDS_BYTES, sa, p); };
    auto store = [&](int Gs) { hipLaunchKernelGGL(store_kernel, dim3(Gs), dim3(1024), 0, sb, d_st, shw, planes, stasks, 1.f); };
    auto run = [&](int G, int Gs, int order, const char* what) {
        // order 0: conv only, 1: store only, 2: store first then conv, 3: conv first then store
        float best_a = 1e9f, best_b = 1e9f, best_t = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            CK(hipStreamWaitEvent(sb, e0, 0));
            if (order == 2) { CK(hipEventRecord(eb0, sb)); store(Gs); CK(hipEventRecord(eb, sb)); }
            if (order != 1) { CK(hipEventRecord(ea0, sa)); conv(G); CK(hipEventRecord(ea, sa)); }
            if (order == 1 || order == 3) { CK(hipEventRecord(eb0, sb)); store(Gs); CK(hipEventRecord(eb, sb)); }
            CK(hipDeviceSynchronize());
            float a = 0, b = 0;
            if (order != 1) CK(hipEventElapsedTime(&a, e0, ea));
            if (order != 0) CK(hipEventElapsedTime(&b, e0, eb));
            const float t = a > b ? a : b;
            if (t < best_t) { best_t = t; best_a = a; best_b = b; }
        }
        printf("%-34s G=%3d Gs=%4d : conv done %.3f ms, store done %.3f ms (%.2f TB/s), makespan %.3f ms\n", what, G, Gs, best_a, best_b,
               best_b > 0 ? planes * shw * 4.0 / best_b * 1e-9 : 0.0, best_t);
        fflush(stdout);
    };
    (void)argc; (void)argv;
    run(256, 0, 0, "conv alone");
    run(224, 0, 0, "conv alone");
    run(192, 0, 0, "conv alone");
    for (int Gs : {16, 32, 64, 128, 256, 1024}) run(0, Gs, 1, "store alone");
    for (int Gs : {16, 32, 64}) {
        run(256 - Gs, Gs, 2, "store first, conv static split");
        run(256, Gs, 2, "store first, conv full grid");
        run(256, Gs, 3, "conv first, then store");
    }
    return 0;
}
