// tools/hidden_bench.hip -- check + timing of spade_hidden_wq (need-masked SPADE hidden activations + one-hot planes for the
// Winograd ACE kernels) against the kernels it replaces (onehot_conv3x3 + label_onehot_planes, every pixel).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hidden_bench.hip -o tools/hidden_bench.bin
#include "../ctrlhair_amd/csrc/sean_kernels.hip"
#include "../ctrlhair_amd/csrc/ace_sparse.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace chk;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    struct Cfg { int B, H, cell, sparse; };
    const Cfg cfgs[] = {{3, 64, 8, 1}, {2, 1024, 64, 1}, {16, 512, 32, 1}, {16, 256, 16, 1}, {16, 128, 8, 1}, {16, 64, 4, 1}, {16, 32, 2, 0}, {16, 512, 32, 0}};
    const int K = 128, KO = 148;
    double tot_old = 0, tot_new = 0;
    int ci = -1;
    for (const Cfg& c : cfgs) {
        ++ci;
        if (only >= 0 && ci != only) continue;
        const int B = c.B, H = c.H, W = c.H, HW = H * W;
        if (!spade_hidden_wq_supported(H, W)) { printf("H=%d skipped\n", H); continue; }
        std::vector<uint8_t> lab((size_t)B * HW);
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    int j = ((x / c.cell) * 7 + (y / c.cell) * 3 + b) % 21;        // labels 19, 20 -> "no class" values 19 and 255
                    lab[((size_t)b * H + y) * W + x] = (uint8_t)(j == 20 ? 255 : j);
                }
        uint8_t *d_lab, *d_u5, *d_need; uint16_t* d_list; int* d_cnt; float *d_tab, *d_bias, *d_ref, *d_out;
        CK(hipMalloc(&d_lab, lab.size())); CK(hipMemcpy(d_lab, lab.data(), lab.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&d_u5, lab.size())); CK(hipMalloc(&d_need, lab.size()));
        const int ntl = B * (W / 32) * (H / 16);
        CK(hipMalloc(&d_list, (size_t)ntl * 512 * 2)); CK(hipMalloc(&d_cnt, ntl * 4));
        CK(ace_classify(d_lab, d_u5, d_need, d_list, d_cnt, B, H, W, 16, 0));
        std::vector<uint8_t> u5(lab.size());
        CK(hipMemcpy(u5.data(), d_u5, u5.size(), hipMemcpyDeviceToHost));
        std::vector<float> tab(19 * 9 * K), bias(K);
        for (auto& v : tab) v = (float)rand() / RAND_MAX - 0.4f;
        for (auto& v : bias) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMalloc(&d_tab, tab.size() * 4)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_bias, K * 4)); CK(hipMemcpy(d_bias, bias.data(), K * 4, hipMemcpyHostToDevice));
        const size_t nout = (size_t)B * KO * HW;
        CK(hipMalloc(&d_ref, nout * 4)); CK(hipMalloc(&d_out, nout * 4));
        CK(hipMemset(d_ref, 0, nout * 4));
        CK(hipMemset(d_out, 0x7F, nout * 4));        // (0x7F7F7F7F: a value no table sum produces)
        CK(onehot_conv3x3(d_lab, d_tab, d_bias, d_ref, B, H, W, K, 1, 0, 0, nullptr, KO));
        CK(label_onehot_planes(d_lab, d_ref, B, H, W, KO, K, 0));
        CK(spade_hidden_wq(d_lab, c.sparse ? d_u5 : nullptr, d_tab, d_bias, d_out, B, H, W, KO, 1, 0));
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> ref(nout), got(nout);
        CK(hipMemcpy(ref.data(), d_ref, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
        // wanted pixels: the 4 x 4 patches of the boundary quads
        std::vector<uint8_t> want((size_t)B * HW, 0);
        for (int b = 0; b < B; ++b)
            for (int qy = 0; qy < H / 2; ++qy)
                for (int qx = 0; qx < W / 2; ++qx) {
                    bool bq = !c.sparse;
                    for (int d = 0; d < 4 && !bq; ++d) bq = u5[((size_t)b * H + 2 * qy + (d >> 1)) * W + 2 * qx + (d & 1)] == 255;
                    if (!bq) continue;
                    for (int dy = -1; dy <= 2; ++dy)
                        for (int dx = -1; dx <= 2; ++dx) {
                            const int y = 2 * qy + dy, x = 2 * qx + dx;
                            if (y >= 0 && y < H && x >= 0 && x < W) want[((size_t)b * H + y) * W + x] = 1;
                        }
                }
        size_t nw = 0, bad = 0, stray = 0;
        for (size_t i = 0; i < want.size(); ++i) nw += want[i];
        // the kernel writes whole aligned groups of 16 pixels: every pixel of a group with a wanted pixel must hold the true value
        std::vector<uint8_t> want16(want.size());
        for (size_t i = 0; i < want.size(); i += 16) {
            uint8_t any = 0;
            for (int e = 0; e < 16; ++e) any |= want[i + e];
            for (int e = 0; e < 16; ++e) want16[i + e] = any;
        }
        size_t nw16 = 0;
        for (size_t i = 0; i < want.size(); ++i) nw16 += want16[i];
        want = want16;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < KO; ++k)
                for (int p = 0; p < HW; ++p) {
                    const size_t i = ((size_t)b * KO + k) * HW + p;
                    if (want[(size_t)b * HW + p]) {
                        uint32_t r = ref[i];
                        if (k == K + 19) r = 0;          // (the zero plane stays zero also under a label value of 19)
                        bad += r != got[i];
                    } else stray += got[i] != 0x7F7F7F7Fu;
                }
        printf("B=%2d H=%3d cell=%2d sparse=%d: wanted %.3f (written %.3f) of the pixels; %zu mismatching words at wanted pixels, %zu words written outside\n", B, H,
               c.cell, c.sparse, (double)nw / want.size(), (double)nw16 / want.size(), bad, stray);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best[2] = {1e9f, 1e9f};
        for (int which = 0; which < 2; ++which)
            for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0, 0));
                if (which == 0) {
                    CK(onehot_conv3x3(d_lab, d_tab, d_bias, d_ref, B, H, W, K, 1, 0, 0, nullptr, KO));
                    CK(label_onehot_planes(d_lab, d_ref, B, H, W, KO, K, 0));
                } else CK(spade_hidden_wq(d_lab, c.sparse ? d_u5 : nullptr, d_tab, d_bias, d_out, B, H, W, KO, 1, 0));
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best[which]) best[which] = ms;
            }
        printf("      old (every pixel) %8.1f us %6.2f TB/s   new %8.1f us %6.2f TB/s of written bytes\n", best[0] * 1e3, (double)B * HW * KO * 4.0 / best[0] * 1e-9,
               best[1] * 1e3, (double)nw16 * KO * 4.0 / best[1] * 1e-9);
        if (B == 16 && c.sparse) { tot_old += best[0] * (H == 512 ? 3 : 3); tot_new += best[1] * 3; }
        hipFree(d_lab); hipFree(d_u5); hipFree(d_need); hipFree(d_list); hipFree(d_cnt); hipFree(d_tab); hipFree(d_bias); hipFree(d_ref); hipFree(d_out);
    }
    printf("three ACEs per level, levels 64 .. 512 (one-hot planes on all of them here): old %.2f ms, new %.2f ms per step\n", tot_old, tot_new);
    return 0;
}
