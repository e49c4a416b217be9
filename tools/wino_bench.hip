// tools/wino_bench.hip -- stand-alone check + timing of the Winograd F(2x2,3x3) exact-f32 MFMA conv (conv_wino.h) against a
// naive direct f32 convolution on the GPU, over the ResBlock conv shapes of the ngf = 64 generator at 512^2, B = 16.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_bench.hip -o tools/wino_bench.bin ; run on the GPU box.
//   wino_bench.bin            all shapes (check + timing)
//   wino_bench.bin quick      small shapes only (check)
#include "../ctrlhair_amd/csrc/conv_inst_wino.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace chk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xFFFF) / 32768.f - 1.f;
}

// direct conv, one thread per output, f32 fma chain in (ci, tap) order, double accumulation option off
__global__ void ref_conv_kernel(const float* in, const float* w, const float* bias, const float* res, int res_up, const float* in2,
                                const float* w2, int Cin2, float* out, int B, int Cin, int Cout, int H, int W) {
    const long long n = (long long)B * Cout * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), co = (int)((i / ((long long)W * H)) % Cout), b = (int)(i / ((long long)W * H * Cout));
        double acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    acc += (double)w[((long long)co * Cin + ci) * 9 + t] * in[(((long long)b * Cin + ci) * H + yy) * W + xx];
            }
        for (int ci = 0; ci < Cin2; ++ci) acc += (double)w2[(long long)co * Cin2 + ci] * in2[(((long long)b * Cin2 + ci) * H + y) * W + x];
        if (res) acc += res[(((long long)b * Cout + co) * (H >> res_up) + (y >> res_up)) * (W >> res_up) + (x >> res_up)];
        out[i] = (float)acc;
    }
}

struct Shape { int B, Cin, Cout, H, Cin2, res; const char* name; };   // res: 0 none, 1 same size, 2 upsampled

__global__ void ref_pw_kernel(const float* in, const float* w, float* out, int B, int Cin, int Cout, int HW) {
    const long long n = (long long)B * Cout * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW), co = (int)((i / HW) % Cout), b = (int)(i / ((long long)HW * Cout));
        double acc = 0;
        for (int ci = 0; ci < Cin; ++ci) acc += (double)w[(long long)co * Cin + ci] * in[((long long)b * Cin + ci) * HW + px];
        out[i] = (float)acc;
    }
}

static unsigned* g_claim = nullptr;      // "dyn" runs: 64 launches x 1024 words (eight counters each), zeroed before every timed series
static int g_claim_i = 0;
static unsigned* next_claim() { return g_claim ? g_claim + 1024 * (g_claim_i++ % 64) : nullptr; }
static int run_pw() {      // the four learned shortcuts of the ngf = 64 generator at 512^2, B = 16 (+ two small shapes)
    struct S1 { int B, Cin, Cout, H; };
    const S1 all[] = {{2, 32, 48, 32}, {1, 64, 32, 16}, {16, 1024, 512, 64}, {16, 512, 256, 128}, {16, 256, 128, 256}, {16, 128, 64, 512}};
    double tot = 0;
    for (const S1& c : all) {
        const int HW = c.H * c.H;
        const size_t nin = (size_t)c.B * c.Cin * HW, nout = (size_t)c.B * c.Cout * HW;
        unsigned seed = 777u + c.Cin;
        std::vector<float> hin(nin), hw((size_t)c.Cout * c.Cin);
        for (auto& v : hin) v = frand(seed);
        for (auto& v : hw) v = frand(seed) / sqrtf((float)c.Cin);
        const float* wp = hw.data();
        const int Cin = c.Cin;
        std::vector<float> pk = pack_pw_A(c.Cout, c.Cin, [&](int row, int ci) { return wp[(size_t)row * Cin + ci]; });
        float *d_in, *d_w, *d_pk, *d_out, *d_ref;
        CK(hipMalloc(&d_in, nin * 4)); CK(hipMalloc(&d_w, hw.size() * 4)); CK(hipMalloc(&d_pk, pk.size() * 4 + 64));
        CK(hipMalloc(&d_out, nout * 4)); CK(hipMalloc(&d_ref, nout * 4));
        CK(hipMemcpy(d_in, hin.data(), nin * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemset(d_out, 0xFF, nout * 4));
        PwParams p{};
        p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = c.B; p.Cin = c.Cin; p.Cout = c.Cout; p.HW = HW;
        if (g_claim) CK(hipMemset(g_claim, 0, 64 * 1024 * 4));
        g_claim_i = 0;
        CK(conv_pw(p, 0));
        hipLaunchKernelGGL(ref_pw_kernel, dim3(4096), dim3(256), 0, 0, d_in, d_w, d_ref, c.B, c.Cin, c.Cout, HW);
        CK(hipDeviceSynchronize());
        std::vector<float> ho(nout), hr(nout);
        CK(hipMemcpy(ho.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), d_ref, nout * 4, hipMemcpyDeviceToHost));
        double maxd = 0;
        for (size_t i = 0; i < nout; ++i) { const double d = fabs((double)ho[i] - hr[i]); if (!(d <= maxd)) maxd = d; }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) CK(conv_pw(p, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        const double fl = 2.0 * c.B * HW * (double)c.Cout * c.Cin;
        printf("1x1  B%2d %4d->%4d %3d^2  maxdiff %.3e %s  %8.3f ms  %6.1f TF/s\n", c.B, c.Cin, c.Cout, c.H, maxd, maxd <= 2e-5 ? "OK  " : "FAIL", ms, fl / ms * 1e-9);
        if (c.B == 16) tot += ms;
        hipFree(d_in); hipFree(d_w); hipFree(d_pk); hipFree(d_out); hipFree(d_ref);
    }
    printf("sum over the four shortcut convs of one step: %.2f ms\n", tot);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strncmp(argv[1], "dyn", 3)) {      // dynamic task claiming of the 3x3 convs
        CK(hipMalloc(&g_claim, 64 * 1024 * 4));
        CK(hipMemset(g_claim, 0, 64 * 1024 * 4));
        --argc;
        ++argv;
    }
    if (argc > 1 && !strcmp(argv[1], "pw")) return run_pw();
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const int dbg = argc > 2 && !strcmp(argv[1], "dbg") ? atoi(argv[2]) : 0;      // timing ablations (wrong results)
    const Shape all[] = {
        {2, 16, 16, 32, 0, 0, "tiny"}, {3, 32, 48, 64, 0, 1, "tiny res, ragged rows"}, {1, 28, 32, 32, 0, 2, "tiny res_up, odd k-steps"},
        {2, 64, 64, 64, 0, 1, "small res"}, {4, 32, 48, 16, 0, 1, "16^2 sample pairs, ragged rows"}, {16, 1024, 1024, 16, 0, 1, "G_head conv (+x), 16^2 pairs"},
        {1, 1024, 1024, 32, 0, 1, "B=1 G_middle conv_1 (+x): split-K"}, {1, 1024, 512, 64, 0, 0, "B=1 up_0 conv_0: split-K"},
        {1, 512, 512, 64, 0, 2, "B=1 512->512 64^2 res_up: split-K"}, {2, 256, 64, 16, 0, 1, "B=2 16^2 pair: split-K"}, {1, 1024, 1024, 16, 0, 1, "B=1 16^2 lone sample in a pair tile: split-K"}, {3, 64, 48, 16, 0, 1, "B=3 16^2: odd batch"},
        {16, 1024, 1024, 32, 0, 0, "G_middle conv_0"}, {16, 1024, 1024, 32, 0, 1, "G_middle conv_1 (+x)"},
        {16, 1024, 512, 64, 0, 0, "up_0 conv_0"}, {16, 512, 512, 64, 0, 1, "up_0 conv_1 (+xs)"},
        {16, 512, 256, 128, 0, 0, "up_1 conv_0"}, {16, 256, 256, 128, 0, 1, "up_1 conv_1 (+xs)"},
        {16, 256, 128, 256, 0, 0, "up_2 conv_0"}, {16, 128, 128, 256, 0, 1, "up_2 conv_1 (+xs)"},
        {16, 128, 64, 512, 0, 0, "up_3 conv_0"}, {16, 64, 64, 512, 0, 1, "up_3 conv_1 (+xs)"},
    };
    float* d_zero;
    CK(hipMalloc(&d_zero, 256));
    CK(hipMemset(d_zero, 0, 256));
    double tot_ms = 0, tot_fl = 0;
    for (const Shape& c : all) {
        if (quick && c.B * (long long)c.H * c.H * c.Cout > (1 << 22)) continue;
        if (dbg && c.B < 16) continue;
        if (!wino_supported(c.H, c.H, c.Cin) && !wino_supported_pair16(c.B, c.H, c.H, c.Cin)) { printf("%-28s skipped (Cin / 4 k-steps do not exceed the ring's run-ahead: direct kernel)\n", c.name); continue; }
        const int B = c.B, Cin = c.Cin, Cout = c.Cout, H = c.H, W = c.H, Cin2 = c.Cin2;
        const size_t nin = (size_t)B * Cin * H * W, nout = (size_t)B * Cout * H * W, nin2 = (size_t)B * Cin2 * H * W;
        const int rh = c.res == 2 ? H / 2 : H;
        const size_t nres = c.res ? (size_t)B * Cout * rh * rh : 0;
        unsigned seed = 12345u + Cin * 7 + Cout;
        std::vector<float> hin(nin), hw((size_t)Cout * Cin * 9), hb(Cout), hin2(nin2), hw2((size_t)Cout * Cin2), hres(nres);
        const float ws = 1.f / sqrtf((float)Cin * 9.f);
        for (auto& v : hin) v = frand(seed);
        for (auto& v : hw) v = frand(seed) * ws;
        for (auto& v : hb) v = frand(seed) * 0.1f;
        for (auto& v : hin2) v = frand(seed);
        for (auto& v : hw2) v = frand(seed) / sqrtf((float)(Cin2 ? Cin2 : 1));
        for (auto& v : hres) v = frand(seed);
        const float* wp = hw.data();
        auto get = [&](int row, int ci, int t) { return wp[((size_t)row * Cin + ci) * 9 + t]; };
        std::vector<float> pk = pack_wino_A(Cout, Cin, get), pk2;
        const float* w2p = hw2.data();
        float *d_in, *d_w, *d_b, *d_pk, *d_out, *d_ref, *d_in2 = nullptr, *d_w2 = nullptr, *d_pk2 = nullptr, *d_res = nullptr;
        CK(hipMalloc(&d_in, nin * 4 + 256)); CK(hipMalloc(&d_w, hw.size() * 4)); CK(hipMalloc(&d_b, Cout * 4));
        CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_out, nout * 4)); CK(hipMalloc(&d_ref, nout * 4));
        CK(hipMemcpy(d_in, hin.data(), nin * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, hb.data(), Cout * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        if (Cin2) {
            CK(hipMalloc(&d_in2, nin2 * 4)); CK(hipMalloc(&d_w2, hw2.size() * 4)); CK(hipMalloc(&d_pk2, pk2.size() * 4));
            CK(hipMemcpy(d_in2, hin2.data(), nin2 * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_pk2, pk2.data(), pk2.size() * 4, hipMemcpyHostToDevice));
        }
        if (nres) {
            CK(hipMalloc(&d_res, nres * 4));
            CK(hipMemcpy(d_res, hres.data(), nres * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemset(d_out, 0xFF, nout * 4));
        WinoParams p{};
        p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
        p.bias = d_b; p.res = d_res; p.res_up = c.res == 2 ? 1 : 0; p.act = ACT_NONE;
        p.zero = d_zero;
        static float* d_part = nullptr;
        if (!d_part) CK(hipMalloc(&d_part, (size_t)(16 << 20) * 4));
        p.partial = getenv("NO_SPLITK") ? nullptr : d_part;
        p.partial_cap = 16 << 20;
        if (g_claim) CK(hipMemset(g_claim, 0, 64 * 1024 * 4));
        g_claim_i = 0;
        p.claim = next_claim();
        CK(conv_wino_plain(p, 0));
        CK(hipDeviceSynchronize());
        // reference (subsampled batch for the big shapes: samples 0 and B-1 only)
        const bool big = (double)nout * Cin * 9 > 4e11;
        const int Bref = big ? 1 : B;
        double maxd = 0, maxr = 0;
        for (int pass = 0; pass < (big ? 2 : 1); ++pass) {
            const int b0 = pass == 0 ? 0 : B - 1;
            hipLaunchKernelGGL(ref_conv_kernel, dim3(4096), dim3(256), 0, 0, d_in + (size_t)b0 * Cin * H * W, d_w, d_b,
                               d_res ? d_res + (size_t)b0 * Cout * rh * rh : nullptr, p.res_up,
                               d_in2 ? d_in2 + (size_t)b0 * Cin2 * H * W : nullptr, d_w2, Cin2, d_ref, Bref, Cin, Cout, H, W);
            CK(hipDeviceSynchronize());
            const size_t nn = (size_t)Bref * Cout * H * W;
            std::vector<float> ho(nn), hr(nn);
            CK(hipMemcpy(ho.data(), d_out + (size_t)b0 * Cout * H * W, nn * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), d_ref, nn * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nn; ++i) {
                const double d = fabs((double)ho[i] - hr[i]);
                if (!(d <= maxd)) maxd = d;      // NaN-propagating
                if (fabs(hr[i]) > maxr) maxr = fabs(hr[i]);
            }
        }
        const double fl = 2.0 * B * H * W * (double)Cout * (Cin * 9.0 + Cin2);
        float ms = 0;
        if (!quick) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int it = 5;
            p.claim = next_claim();
            CK(conv_wino_plain(p, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < it; ++i) { p.claim = next_claim(); CK(conv_wino_plain(p, 0)); }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= it;
            tot_ms += ms * (strstr(c.name, "G_middle") ? 2 : 1);
            tot_fl += fl * (strstr(c.name, "G_middle") ? 2 : 1);
        }
        const double exec = 2.0 * B * (H / 2) * (W / 2) * (double)Cout * (Cin * 16.0 + Cin2 * 4.0);
        printf("%-28s B%2d %4d->%4d (+%4d) %3d^2  maxdiff %.3e (max|ref| %.2f)  %s  %8.3f ms  dense %6.1f TF/s  executed %6.1f TF/s\n", c.name, B,
               Cin, Cout, Cin2, H, maxd, maxr, maxd <= 2e-5 * (maxr > 1 ? maxr : 1) ? "OK  " : "FAIL", ms, ms > 0 ? fl / ms * 1e-9 : 0.0,
               ms > 0 ? exec / ms * 1e-9 : 0.0);
        fflush(stdout);
        hipFree(d_in); hipFree(d_w); hipFree(d_b); hipFree(d_pk); hipFree(d_out); hipFree(d_ref);
        if (d_in2) { hipFree(d_in2); hipFree(d_w2); hipFree(d_pk2); }
        if (d_res) hipFree(d_res);
    }
    if (!quick) printf("sum over the ResBlock convs of one step (G_middle x2): %.2f ms, dense-equivalent %.1f TF/s\n", tot_ms, tot_fl / tot_ms * 1e-9);
    return 0;
}
