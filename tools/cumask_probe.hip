// tools/cumask_probe.hip -- does hipExtStreamCreateWithCUMask confine a WIDE store-bound kernel to a few CUs on gfx950 (8 XCCs), and
// does the product's persistent Winograd conv (full mask, other stream) run under it?  Census of (XCC, SE, CU) ids per masked stream.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cumask_probe.hip -o tools/cumask_probe.bin
#include "../ctrlhair_amd/csrc/conv_inst_wino.hip"
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

using namespace chk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// the label-table kernel's store pattern: one thread = one pixel, 4-byte stores to 32 planes 'hw' apart
__global__ __launch_bounds__(256) void store4_kernel(float* out, long long hw, float v) {
    const long long pix = blockIdx.x * 256ll + threadIdx.x;
    float* base = out + (long long)blockIdx.y * 32 * hw + pix;
#pragma unroll 8
    for (int c = 0; c < 32; ++c) base[c * hw] = v + c;
}
__global__ void census_kernel(unsigned* ids) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // XCC_ID
        ids[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFF00) | ((hw >> 13) & 0x7) << 4;      // cu_id bits 11:8, sh 12, se 15:13
        for (volatile int i = 0; i < 20000; ++i) {}
    }
}

int main() {
    const int B = 16, Cin = 128, Cout = 128, H = 256, W = 256;
    const size_t nin = (size_t)B * Cin * H * W, nout = (size_t)B * Cout * H * W;
    std::vector<float> hin(nin), hw_((size_t)Cout * Cin * 9);
    unsigned s = 1234;
    auto fr = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& v : hin) v = fr();
    for (auto& v : hw_) v = fr() / 34.f;
    const float* wp = hw_.data();
    std::vector<float> pk = pack_wino_A(Cout, Cin, [&](int row, int ci, int t) { return wp[((size_t)row * Cin + ci) * 9 + t]; });
    float *d_in, *d_pk, *d_out, *d_zero, *d_st;
    CK(hipMalloc(&d_in, nin * 4 + 256)); CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_out, nout * 4)); CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    CK(hipMemcpy(d_in, hin.data(), nin * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    const long long shw = 16ll * 512 * 512;
    const int planes = 128;
    CK(hipMalloc(&d_st, (size_t)planes * shw * 4));
    unsigned* d_ids;
    CK(hipMalloc(&d_ids, 8192 * 4));
    WinoParams p{};
    p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W; p.act = ACT_NONE; p.zero = d_zero;
    wino_fill_launch(p);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_plain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino::LDS_BYTES));
    hipStream_t sa;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    hipEvent_t e0, ea, eb;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    const int NC = 4;
    auto conv = [&](int G) { for (int i = 0; i < NC; ++i) hipLaunchKernelGGL(wino_plain_kernel<0>, dim3(G), dim3(512), wino::LDS_BYTES, sa, p); };
    for (int ncu : {256, 16, 32, 48, 64}) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
        hipStream_t sb;
        hipError_t e = hipExtStreamCreateWithCUMask(&sb, 8, mask);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%d CUs) failed: %s\n", ncu, hipGetErrorString(e)); continue; }
        CK(hipMemset(d_ids, 0xFF, 8192 * 4));
        hipLaunchKernelGGL(census_kernel, dim3(4096), dim3(64), 0, sb, d_ids);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> ids(4096);
        CK(hipMemcpy(ids.data(), d_ids, 4096 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus, xccs;
        for (unsigned v : ids) { cus.insert(v); xccs.insert(v >> 16); }
        printf("mask of %3d CUs: census sees %zu distinct (xcc, se, cu) on %zu XCCs\n", ncu, cus.size(), xccs.size());
        auto store = [&]() { hipLaunchKernelGGL(store4_kernel, dim3((unsigned)(shw / 256), planes / 32), dim3(256), 0, sb, d_st, shw, 1.f); };
        for (int order = 1; order <= 3; ++order) {      // 1 store alone, 2 store first + conv, 3 conv first + store
            float best_a = 0, best_b = 0, best_t = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, sa));
                CK(hipStreamWaitEvent(sb, e0, 0));
                if (order == 2) { store(); CK(hipEventRecord(eb, sb)); }
                if (order != 1) { conv(256); CK(hipEventRecord(ea, sa)); }
                if (order != 2) { store(); CK(hipEventRecord(eb, sb)); }
                CK(hipDeviceSynchronize());
                float a = 0, b = 0;
                if (order != 1) CK(hipEventElapsedTime(&a, e0, ea));
                CK(hipEventElapsedTime(&b, e0, eb));
                const float t = a > b ? a : b;
                if (t < best_t) { best_t = t; best_a = a; best_b = b; }
            }
            printf("   %-22s conv done %.3f ms, store done %.3f ms (%.2f TB/s), makespan %.3f\n",
                   order == 1 ? "store alone" : (order == 2 ? "store first + 4 convs" : "4 convs first + store"), best_a, best_b,
                   planes * shw * 4.0 / best_b * 1e-9, best_t);
        }
        fflush(stdout);
        CK(hipStreamDestroy(sb));
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa)); conv(256); CK(hipEventRecord(ea, sa)); CK(hipDeviceSynchronize());
    float a; CK(hipEventElapsedTime(&a, e0, ea));
    printf("4 convs alone: %.3f ms\n", a);
    return 0;
}
