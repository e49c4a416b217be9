// tools/w4w_timeline.hip -- s_memtime stamps inside the wide-tile F(4x4,3x3) kernel (tools/rejected/conv_wino4w.h, one wave per SIMD): where
// the cycles of a k-step go, group by group.  Timing only (the stamps disturb the compiler's wait counting: results are not checked).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DW4W_STAMP tools/w4w_timeline.hip -o tools/w4w_timeline.bin
#include "../ctrlhair_amd/csrc/conv_inst_wino4.hip"
#include "rejected/conv_wino4w.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace chk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int B = 16, Cin = 1024, Cout = 1024, H = 32, W = 32;
    std::vector<float> hin((size_t)B * Cin * H * W), hw((size_t)Cout * Cin * 9);
    unsigned s = 1;
    auto fr = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.f - 1.f; };
    for (auto& v : hin) v = fr();
    for (auto& v : hw) v = fr() * 0.02f;
    std::vector<float> pk = pack_wino4_A(Cout, Cin, [&](int r, int ci, int t) { return r < Cout ? hw[((size_t)r * Cin + ci) * 9 + t] : 0.f; });
    float *d_in, *d_pk, *d_out;
    CK(hipMalloc(&d_in, hin.size() * 4)); CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_out, (size_t)B * Cout * H * W * 4));
    CK(hipMemcpy(d_in, hin.data(), hin.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    Wino4Params p{};
    p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    wino4_fill_launch(p);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_plain_w_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4::LDS_BYTES));
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(wino4_plain_w_kernel<0>, dim3(256), dim3(256), wino4::LDS_BYTES, 0, p);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> st(4 * 64 * 12);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(w4w_stamps), st.size() * 8));
    for (int w = 0; w < 4; ++w) {
        printf("wave %d: per k-step [wait+barrier | groups 0..8 | total] in cycles (s_memtime ticks)\n", w);
        for (int k = 8; k < 24; ++k) {
            const unsigned long long* t = &st[(w * 64 + k) * 12];
            printf("  k%2d: %5llu |", k, t[1] - t[0]);
            for (int g = 0; g < 9; ++g) printf(" %4llu", t[2 + g] - t[1 + g]);
            const unsigned long long* tn = &st[(w * 64 + k + 1) * 12];
            printf(" | %5llu  (to next top %5llu)\n", t[10] - t[0], tn[0] - t[0]);
        }
    }
    return 0;
}
