// tools/wino4_bench.hip -- stand-alone check + timing of the Winograd F(4x4,3x3) exact-f32 MFMA conv (conv_wino4.h) against a naive
// direct convolution on the GPU (double accumulation), over the ResBlock conv shapes of the ngf = 64 generator at 512^2, B = 16.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino4_bench.hip -o tools/wino4_bench.bin ; run on the GPU box.
#include "../ctrlhair_amd/csrc/conv_inst_wino4.hip"
#ifdef W4S_EXPERIMENT      // the shared-transform variant (tools/wino4s_experiment.h): measured, not adopted
#include "wino4s_experiment.h"
namespace chk {
static hipError_t conv_wino4s_plain(Wino4Params p, hipStream_t s) {
    if (!wino4_supported(p.H, p.W, p.Cin) || !p.in || !p.wpk || !p.out) return hipErrorInvalidValue;
    wino4_fill_launch(p);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4s_plain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4s::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4s_plain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4s::LDS_BYTES);
    if (e != hipSuccess) return e;
    const int grid = p.ntasks < 256 ? p.ntasks : 256;
    if (p.reflect) hipLaunchKernelGGL(wino4s_plain_kernel<1>, dim3(grid), dim3(512), wino4s::LDS_BYTES, s, p);
    else hipLaunchKernelGGL(wino4s_plain_kernel<0>, dim3(grid), dim3(512), wino4s::LDS_BYTES, s, p);
    return hipGetLastError();
}
}  // namespace chk
#endif
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
__global__ void ref_conv_kernel(const float* in, const float* w, const float* bias, const float* res, int res_up, float* out, int B, int Cin, int Cout,
                                int H, int W, int refl) {
    const long long n = (long long)B * Cout * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), co = (int)((i / ((long long)W * H)) % Cout), b = (int)(i / ((long long)W * H * Cout));
        double acc = bias ? bias[co] : 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (refl) {
                    yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                    xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                }
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                    acc += (double)w[((long long)co * Cin + ci) * 9 + t] * in[(((long long)b * Cin + ci) * H + yy) * W + xx];
            }
        if (res) acc += res[(((long long)b * Cout + co) * (H >> res_up) + (y >> res_up)) * (W >> res_up) + (x >> res_up)];
        out[i] = refl ? (float)tanh(acc) : (float)acc;
    }
}
struct Shape { int B, Cin, Cout, H, res; const char* name; };   // res: 0 none, 1 same size, 2 upsampled, 3 = reflection padding + tanh

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    const Shape all[] = {
        {2, 16, 16, 32, 0, "tiny"}, {3, 32, 48, 64, 1, "tiny res, ragged rows"}, {1, 24, 32, 32, 2, "tiny res_up"}, {2, 64, 64, 96, 1, "96^2 res"},
        {3, 16, 40, 64, 3, "tiny reflect + tanh"}, {8, 256, 512, 256, 3, "Zencoder 256->512 reflect + tanh"},
        {16, 1024, 1024, 32, 0, "G_middle conv_0"}, {16, 1024, 1024, 32, 1, "G_middle conv_1 (+x)"},
        {16, 1024, 512, 64, 0, "up_0 conv_0"}, {16, 512, 512, 64, 1, "up_0 conv_1 (+xs)"},
        {16, 512, 256, 128, 0, "up_1 conv_0"}, {16, 256, 256, 128, 1, "up_1 conv_1 (+xs)"},
        {16, 256, 128, 256, 0, "up_2 conv_0"}, {16, 128, 128, 256, 1, "up_2 conv_1 (+xs)"},
        {16, 128, 64, 512, 0, "up_3 conv_0"}, {16, 64, 64, 512, 1, "up_3 conv_1 (+xs)"},
    };
    double tot_ms = 0, tot_fl = 0, tot_ms_s = 0;
    for (const Shape& c : all) {
        if (quick && c.B * (long long)c.H * c.H * c.Cout > (1 << 22)) continue;
        const int B = c.B, Cin = c.Cin, Cout = c.Cout, H = c.H, W = c.H;
        if (!wino4_supported(H, W, Cin)) { printf("%-28s not supported\n", c.name); continue; }
        const size_t nin = (size_t)B * Cin * H * W, nout = (size_t)B * Cout * H * W;
        const int rh = c.res == 2 ? H / 2 : H;
        const size_t nres = c.res ? (size_t)B * Cout * rh * rh : 0;
        unsigned seed = 12345u + Cin * 7 + Cout;
        std::vector<float> hin(nin), hw((size_t)Cout * Cin * 9), hb(Cout), hres(nres);
        const float ws = 1.f / sqrtf((float)Cin * 9.f);
        for (auto& v : hin) v = frand(seed);
        for (auto& v : hw) v = frand(seed) * ws;
        for (auto& v : hb) v = frand(seed) * 0.1f;
        for (auto& v : hres) v = frand(seed);
        const float* wp = hw.data();
        std::vector<float> pk = pack_wino4_A(Cout, Cin, [&](int row, int ci, int t) { return wp[((size_t)row * Cin + ci) * 9 + t]; });
        float *d_in, *d_w, *d_b, *d_pk, *d_out, *d_ref, *d_res = nullptr;
        CK(hipMalloc(&d_in, nin * 4 + 256)); CK(hipMalloc(&d_w, hw.size() * 4)); CK(hipMalloc(&d_b, Cout * 4));
        CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMalloc(&d_out, nout * 4)); CK(hipMalloc(&d_ref, nout * 4));
        CK(hipMemcpy(d_in, hin.data(), nin * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_b, hb.data(), Cout * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        if (nres) { CK(hipMalloc(&d_res, nres * 4)); CK(hipMemcpy(d_res, hres.data(), nres * 4, hipMemcpyHostToDevice)); }
        CK(hipMemset(d_out, 0xFF, nout * 4));
        Wino4Params p{};
        p.in = d_in; p.wpk = d_pk; p.out = d_out; p.B = B; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
        p.bias = d_b; p.res = c.res == 3 ? nullptr : d_res; p.res_up = c.res == 2 ? 1 : 0;
        p.reflect = c.res == 3; p.act = c.res == 3 ? ACT_TANH : ACT_NONE;
        CK(conv_wino4_plain(p, 0));
        CK(hipDeviceSynchronize());
        const bool big = (double)nout * Cin * 9 > 4e11;
        const int Bref = big ? 1 : B;
        double maxd = 0, maxr = 0;
        for (int pass = 0; pass < (big ? 2 : 1); ++pass) {
            const int b0 = pass == 0 ? 0 : B - 1;
            hipLaunchKernelGGL(ref_conv_kernel, dim3(4096), dim3(256), 0, 0, d_in + (size_t)b0 * Cin * H * W, d_w, d_b,
                               (d_res && c.res != 3) ? d_res + (size_t)b0 * Cout * rh * rh : nullptr, p.res_up, d_ref, Bref, Cin, Cout, H, W, c.res == 3 ? 1 : 0);
            CK(hipDeviceSynchronize());
            const size_t nn = (size_t)Bref * Cout * H * W;
            std::vector<float> ho(nn), hr(nn);
            CK(hipMemcpy(ho.data(), d_out + (size_t)b0 * Cout * H * W, nn * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), d_ref, nn * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nn; ++i) {
                const double d = fabs((double)ho[i] - hr[i]);
                if (!(d <= maxd)) maxd = d;
                if (fabs(hr[i]) > maxr) maxr = fabs(hr[i]);
            }
        }
        const double fl = 2.0 * B * H * W * (double)Cout * Cin * 9.0;
        float ms = 0;
        if (!quick) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            const int it = 5;
            CK(conv_wino4_plain(p, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < it; ++i) CK(conv_wino4_plain(p, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= it;
            if (B == 16) { tot_ms += ms * (strstr(c.name, "G_middle") ? 2 : 1); tot_fl += fl * (strstr(c.name, "G_middle") ? 2 : 1); }
        }
#ifdef W4S_EXPERIMENT
        // one input transform per pair of waves (tools/wino4s_experiment.h): must equal the result above bit for bit
        {
            float* d_out3;
            CK(hipMalloc(&d_out3, nout * 4));
            CK(hipMemset(d_out3, 0xFF, nout * 4));
            Wino4Params ps = p;
            ps.out = d_out3;
            CK(conv_wino4s_plain(ps, 0));
            CK(hipDeviceSynchronize());
            std::vector<float> h1(nout), h3(nout);
            CK(hipMemcpy(h1.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h3.data(), d_out3, nout * 4, hipMemcpyDeviceToHost));
            const bool same = memcmp(h1.data(), h3.data(), nout * 4) == 0;
            double sd = 0;
            if (!same) for (size_t i = 0; i < nout; ++i) { const double d = fabs((double)h1[i] - h3[i]); if (!(d <= sd)) sd = d; }
            float mss = 0;
            if (!quick) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                const int it = 5;
                CK(conv_wino4s_plain(ps, 0));
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < it; ++i) CK(conv_wino4s_plain(ps, 0));
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&mss, e0, e1));
                mss /= it;
                if (B == 16) tot_ms_s += mss * (strstr(c.name, "G_middle") ? 2 : 1);
            }
            printf("   shared transform: %s (max diff %.3e)  %8.3f ms  executed %6.1f TF/s\n", same ? "BITEQ" : "DIFF ", sd, mss,
                   mss > 0 ? 2.0 * B * (H / 4) * (W / 4) * (double)Cout * Cin * 36.0 / mss * 1e-9 : 0.0);
            (void)hipFree(d_out3);
        }
#endif
        // pre-transformed input route (conv_wino4v.h): pack pass + contraction; must equal the in-kernel-transform result bit for bit
        float msv = 0, msp = 0;
        double vdiff = -1;
        if (Cin % 8 == 0) {
            float *d_v, *d_out2;
            const size_t vb = wino4v_bytes(B, H, W, Cin / 4);
            CK(hipMalloc(&d_v, vb)); CK(hipMalloc(&d_out2, nout * 4));
            CK(hipMemset(d_out2, 0xFF, nout * 4));
            Wino4vPackParams pp{};
            pp.in = d_in; pp.v = d_v; pp.B = B; pp.K = Cin; pp.H = H; pp.W = W; pp.nks = Cin / 4; pp.pitch = W; pp.xoff = 0; pp.padded = 0; pp.reflect = p.reflect;
            Wino4Params pv = p;
            pv.v = d_v; pv.out = d_out2;
            CK(wino4v_pack(pp, 0));
            CK(conv_wino4v_plain(pv, 0));
            CK(hipDeviceSynchronize());
            std::vector<float> h1(nout), h2(nout);
            CK(hipMemcpy(h1.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), d_out2, nout * 4, hipMemcpyDeviceToHost));
            vdiff = 0;
            for (size_t i = 0; i < nout; ++i) { const double d = fabs((double)h1[i] - h2[i]); if (!(d <= vdiff)) vdiff = d; }
            if (!quick) {
                hipEvent_t e0, e1, e2;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
                const int it = 5;
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < it; ++i) CK(wino4v_pack(pp, 0));
                CK(hipEventRecord(e1, 0));
                for (int i = 0; i < it; ++i) CK(conv_wino4v_plain(pv, 0));
                CK(hipEventRecord(e2, 0));
                CK(hipEventSynchronize(e2));
                CK(hipEventElapsedTime(&msp, e0, e1)); CK(hipEventElapsedTime(&msv, e1, e2));
                msp /= it; msv /= it;
            }
            (void)hipFree(d_v); (void)hipFree(d_out2);
        }
        const double exec = 2.0 * B * (H / 4) * (W / 4) * (double)Cout * Cin * 36.0;
        printf("   V route: max|v - in-kernel| %.3e %s  pack %7.3f ms + conv %7.3f ms = %7.3f ms  (conv alone executed %6.1f TF/s)\n", vdiff,
               vdiff == 0 ? "BITEQ" : "DIFF ", msp, msv, msp + msv, msv > 0 ? 2.0 * B * (H / 4) * (W / 4) * (double)Cout * Cin * 36.0 / msv * 1e-9 : 0.0);
        printf("%-28s B%2d %4d->%4d %3d^2  maxdiff %.3e (max|ref| %.2f)  %s  %8.3f ms  dense %6.1f TF/s  executed %6.1f TF/s\n", c.name, B, Cin, Cout, H, maxd,
               maxr, maxd <= 2e-4 * (maxr > 1 ? maxr : 1) ? "OK  " : "FAIL", ms, ms > 0 ? fl / ms * 1e-9 : 0.0, ms > 0 ? exec / ms * 1e-9 : 0.0);
        fflush(stdout);
        (void)hipFree(d_in); (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_pk); (void)hipFree(d_out); (void)hipFree(d_ref);
        if (d_res) (void)hipFree(d_res);
    }
    if (!quick) printf("sum over the ResBlock convs of one step from 32^2 up (G_middle x2): %.2f ms, dense-equivalent %.1f TF/s; shared transform: %.2f ms\n", tot_ms, tot_fl / tot_ms * 1e-9, tot_ms_s);
    return 0;
}
