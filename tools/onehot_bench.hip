// tools/onehot_bench.hip -- stand-alone timing of the need-masked label-table kernel of the f16x3 path
// (onehot_conv3x3_sh16: persistent 32 x 8-pixel blocks vs one block per 256 pixels).  Build like tools/interior_bench.hip.
#include "../ctrlhair_amd/csrc/sean_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace chk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main() {
    struct Cfg { int B, H, cell; };
    const Cfg cfgs[] = {{16, 512, 64}, {16, 512, 32}, {16, 256, 32}, {16, 256, 16}, {16, 128, 16}, {16, 64, 8}};
    const int K = 128;
    for (const Cfg& c : cfgs) {
        const int B = c.B, H = c.H, W = c.H, HW = H * W;
        std::vector<uint8_t> lab((size_t)B * HW), need((size_t)B * HW);
        size_t nneed = 0;
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    lab[((size_t)b * H + y) * W + x] = (uint8_t)(((x / c.cell) + (y / c.cell) * 3 + b) % 19);
                    const int mx = x % c.cell, my = y % c.cell;      // boundary band 2 + 2 pixels at every cell edge, need = band + 1
                    const bool nd = mx < 3 || mx >= c.cell - 3 || my < 3 || my >= c.cell - 3;
                    need[((size_t)b * H + y) * W + x] = nd;
                    nneed += nd;
                }
        uint8_t *d_lab, *d_need; float *d_tab, *d_bias; void* d_out;
        CK(hipMalloc(&d_lab, lab.size())); CK(hipMemcpy(d_lab, lab.data(), lab.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_need, need.size())); CK(hipMemcpy(d_need, need.data(), need.size(), hipMemcpyHostToDevice));
        std::vector<float> tab(19 * 9 * K), bias(K);
        for (auto& v : tab) v = (float)rand() / RAND_MAX - 0.4f;
        for (auto& v : bias) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_bias, K * 4)); CK(hipMemcpy(d_bias, bias.data(), K * 4, hipMemcpyHostToDevice));
        const size_t nout = (size_t)B * K * HW;                     // SH16: 4 bytes per element
        CK(hipMalloc(&d_out, nout * 4));
        std::vector<uint32_t> ref(nout), got(nout);
        CK(hipMemset(d_out, 0, nout * 4));
        CK(onehot_conv3x3_sh16(d_lab, d_tab, d_bias, d_out, B, H, W, K, 1, 64.f, 0, 0, d_need, nullptr, 1));
        CK(hipMemcpy(ref.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(d_out, 0, nout * 4));
        CK(onehot_conv3x3_sh16(d_lab, d_tab, d_bias, d_out, B, H, W, K, 1, 64.f, 0, 0, d_need, nullptr, 2));
        CK(hipMemcpy(got.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < nout; ++i) bad += ref[i] != got[i];
        printf("  check: %zu mismatching words\n", bad);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int which = 0; which < 3; ++which) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0, 0));
                CK(onehot_conv3x3_sh16(d_lab, d_tab, d_bias, d_out, B, H, W, K, 1, 64.f, 0, 0, which == 2 ? nullptr : d_need, nullptr, which == 0 ? 1 : 2));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            const double bytes = (which == 2 ? (double)B * HW : (double)nneed) * K * 4.0;
            printf("%s B=%d H=%3d cell=%2d need=%.2f : %8.1f us  %7.1f GB/s written\n",
                   which == 0 ? "per-256-px blocks" : (which == 1 ? "compacting 32x32  " : "dense (no need)   "), B, H, c.cell,
                   (double)nneed / ((double)B * HW), best * 1e3, bytes / best * 1e-6);
        }
        hipFree(d_lab); hipFree(d_need); hipFree(d_tab); hipFree(d_bias); hipFree(d_out);
    }
    return 0;
}
