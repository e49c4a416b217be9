// tools/wino_ace_bench.hip -- stand-alone timing of wino_ace_kernel (conv_wino.h: SPADE gamma/beta conv + style k-steps + fused ACE
// epilogue over the boundary quads) on the ACE shapes of the ngf = 64 generator at 512^2, B = 16, blocky benchmark labels.
// Operands are random: this tool measures, the parity tests live in tests/test_hip_wino.py and the goldens.  (The cycle stamps and
// timing ablations of the first version of the kernel -- whose findings DESIGN.md section 7 records -- were compiled into the
// kernel itself and slowed it by 5-25 %: the product kernel carries no instrumentation.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_ace_bench.hip -o tools/wino_ace_bench.bin ; run on the GPU box.
//   wino_ace_bench.bin [mode: 0 = tile kernel, 1 = gather kernel] [TH override of the tile kernel: 0 = by level] [only r]
#include "../ctrlhair_amd/csrc/conv_inst_wino.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace chk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static float frand(unsigned& s) { return (rnd(s) & 0xFFFF) / 32768.f - 1.f; }

__global__ void fill_kernel(float* p, long long n, unsigned seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned s = seed + (unsigned)i * 2654435761u;
        s = s * 1664525u + 1013904223u;
        p[i] = ((s >> 8) & 0xFFFF) / 32768.f - 1.f;
    }
}
static float* dev_rand(long long n, unsigned seed) {
    float* p;
    CK(hipMalloc(&p, n * 4 + 256));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, seed);
    return p;
}

struct Shape { int r, C, styled, x_up; const char* name; };

int main(int argc, char** argv) {
    const int gather = argc > 1 ? atoi(argv[1]) : 0;
    const int th_over = argc > 2 ? atoi(argv[2]) : 0;
    const int only_r = argc > 3 ? atoi(argv[3]) : 0;
    const int dyn = argc > 4 ? atoi(argv[4]) : 0;      // 1 = dynamic task claiming (gather kernel)
    unsigned* d_claim;
    CK(hipMalloc(&d_claim, 64 * 1024 * 4));      // per launch: 8 counters, pad, 2 mailbox words per block
    const int B = 16, grid = 16;
    const Shape all[] = {
        {64, 1024, 1, 1, "up_0 ace_s/ace_0"}, {64, 512, 1, 0, "up_0 ace_1"},
        {128, 512, 1, 1, "up_1 ace_s/ace_0"}, {128, 256, 1, 0, "up_1 ace_1"},
        {256, 256, 1, 1, "up_2 ace_s/ace_0"}, {256, 128, 1, 0, "up_2 ace_1"},
        {512, 128, 0, 1, "up_3 ace_s/ace_0"}, {512, 64, 0, 0, "up_3 ace_1"},
    };
    float* d_zero;
    CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    double tot = 0;
    for (const Shape& c : all) {
        if (only_r && c.r != only_r) continue;
        const int r = c.r, C = c.C, K = 128 + (c.styled ? 20 : 0), nrt = (C + 15) / 16;
        const int TH = gather ? 16 : (th_over ? th_over : ((r >= 512 || r == 128) ? 32 : 16));
        // blocky labels (grid x grid blocks per sample) and the interior map: 5x5 uniform and two pixels inside the image
        unsigned seed = 4242u + r;
        std::vector<uint8_t> lab((size_t)B * r * r), u5((size_t)B * r * r);
        const int rep = r / grid;
        for (int b = 0; b < B; ++b) {
            uint8_t g[16][16];
            for (auto& row : g) for (auto& v : row) v = (uint8_t)(rnd(seed) % 19);
            for (int y = 0; y < r; ++y) for (int x = 0; x < r; ++x) lab[((size_t)b * r + y) * r + x] = g[y / rep][x / rep];
        }
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < r; ++y)
                for (int x = 0; x < r; ++x) {
                    const uint8_t l = lab[((size_t)b * r + y) * r + x];
                    bool uni = y >= 2 && x >= 2 && y < r - 2 && x < r - 2;
                    for (int dy = -2; dy <= 2 && uni; ++dy)
                        for (int dx = -2; dx <= 2 && uni; ++dx) uni = lab[((size_t)b * r + y + dy) * r + x + dx] == l;
                    u5[((size_t)b * r + y) * r + x] = uni ? l : 255;
                }
        uint8_t* d_u5;
        CK(hipMalloc(&d_u5, u5.size()));
        CK(hipMemcpy(d_u5, u5.data(), u5.size(), hipMemcpyHostToDevice));
        const int ntiles = B * (r / 32) * (r / TH);
        uint8_t* d_ql; int *d_qc, *d_pc, *d_tot; unsigned* d_work;
        CK(hipMalloc(&d_ql, (size_t)ntiles * 8 * TH)); CK(hipMalloc(&d_qc, ntiles * 4)); CK(hipMalloc(&d_pc, ntiles * 4));
        CK(hipMalloc(&d_tot, 32)); CK(hipMalloc(&d_work, ((size_t)ntiles * 4 + B) * ((nrt + 1) / 2) * 4));
        CK(wino_quad_lists(r >= 64 ? d_u5 : nullptr, d_ql, d_qc, d_pc, B, r, r, TH, 0));
        unsigned* d_gq = nullptr; int *d_gqn = nullptr, *d_qoff = nullptr;
        const int gq_cap = (r / 2) * (r / 2);
        if (gather) {
            CK(hipMalloc(&d_gq, (size_t)B * gq_cap * 4)); CK(hipMalloc(&d_gqn, 32 * 4)); CK(hipMalloc(&d_qoff, ntiles * 4));
            CK(wino_gather_lists(d_ql, d_qc, d_qoff, d_gq, d_gqn, gq_cap, B, r, r, 0));
            CK(wino_gather_worklist(d_gqn, d_pc, B, ntiles / B, nrt, d_work, d_tot, 0));
        } else
            CK(wino_ace_worklist(d_qc, d_pc, ntiles, nrt, d_work, d_tot, 0));
        int tot_h[8];
        CK(hipMemcpy(tot_h, d_tot, 32, hipMemcpyDeviceToHost));
        const long long px = (long long)B * r * r, xpx = c.x_up ? px / 4 : px;
        float* d_actv = dev_rand((long long)B * r * wino_apitch(r) * K, 1);      // padded planes (conv_wino.h WINO_AXOFF)
        float* d_wpk = dev_rand((long long)nrt * 32 * 2048, 2);
        float* d_wsty = c.styled ? dev_rand((long long)B * nrt * 5 * 2048, 3) : nullptr;
        float* d_x = dev_rand(xpx * C, 4);
        float* d_out = dev_rand(px * C, 5);
        float* d_noise = dev_rand(px, 6);
        float* d_par = dev_rand(5 * C, 7);
        WinoAceParams w{};
        w.actv = d_actv; w.wpk = d_wpk; w.wsty = d_wsty; w.out = d_out; w.x = d_x; w.x_up = c.x_up; w.act = ACT_LRELU;
        w.B = B; w.C = C; w.H = r; w.W = r;
        w.bias_g = d_par; w.bias_b = d_par + C; w.bn_a = d_par + 2 * C; w.bn_d = d_par + 3 * C; w.nv = d_par + 4 * C;
        w.noise = d_noise; w.noise_bstride = (long long)r * r;
        w.qlist = d_ql; w.TH = TH; w.qcnt = d_qc; w.work = d_work; w.total = d_tot; w.zero = d_zero;
        w.gq = d_gq; w.gq_n = d_gqn; w.gq_cap = gq_cap;
        CK(hipMemset(d_claim, 0, 64 * 1024 * 4));
        w.claim = dyn ? d_claim : nullptr;
        CK(conv_wino_ace(w, 0));
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int it = 5;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < it; ++i) {
            w.claim = dyn ? d_claim + 1024 * (i + 1) : nullptr;
            CK(conv_wino_ace(w, 0));
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= it;
        const double fl = (double)tot_h[3] * 32 * 16 * 16 * K * 2.0;          // wave tasks x 32 rows x 16 quad slots x 16 positions x K
        const double fl_q = (double)tot_h[1] * 2 * C * 16 * K * 2.0;          // boundary quads only (no padding of the 16-quad groups)
        printf("%-18s r %3d C %4d K %3d TH %2d  tasks %6d (%.1f / CU) quads %7d fill %.3f  %7.3f ms  %6.1f TF/s on slots, %6.1f on quads\n",
               c.name, r, C, K, TH, tot_h[0], tot_h[0] / 256.0, tot_h[1], tot_h[1] / (16.0 * tot_h[2]), ms, fl / ms * 1e-9, fl_q / ms * 1e-9);
        fflush(stdout);
        tot += ms * (c.x_up ? 2 : 1);
        if (d_gq) { hipFree(d_gq); hipFree(d_gqn); hipFree(d_qoff); }
        hipFree(d_u5); hipFree(d_ql); hipFree(d_qc); hipFree(d_pc); hipFree(d_tot); hipFree(d_work);
        hipFree(d_actv); hipFree(d_wpk); if (d_wsty) hipFree(d_wsty); hipFree(d_x); hipFree(d_out); hipFree(d_noise); hipFree(d_par);
    }
    printf("sum over the 12 Winograd ACE launches of one step: %.2f ms\n", tot);
    return 0;
}
