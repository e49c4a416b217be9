// net_common.h -- shared host-side helpers of the model files: host tensor store, weight upload, MFMA operand
// packing, a generic conv layer (conv_mfma_kernel launcher wrapper) and small-op launch helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "conv_mfma.h"
#include "conv_wino.h"

namespace chk {

struct HostTensor {
    std::vector<char> data;
    std::vector<int64_t> shape;
    int dtype = 0;  // 0 f32, 1 i64
    const float* f32() const { return reinterpret_cast<const float*>(data.data()); }
    size_t numel() const {
        size_t n = 1;
        for (auto d : shape) n *= (size_t)d;
        return n;
    }
};
typedef std::map<std::string, HostTensor> TensorStore;

struct Builder {
    const TensorStore& ts;
    std::vector<void*>& allocs;
    std::string err;
    std::string prefix;   // prepended to every tensor name looked up
    Builder(const TensorStore& t, std::vector<void*>& a) : ts(t), allocs(a) {}

    bool has(const std::string& n) const { return ts.find(prefix + n) != ts.end(); }
    const HostTensor* get(const std::string& n0, size_t numel) {
        const std::string n = prefix + n0;
        auto it = ts.find(n);
        if (it == ts.end()) {
            if (err.empty()) err = "missing tensor '" + n + "'";
            return nullptr;
        }
        if (it->second.dtype != 0 || it->second.numel() != numel) {
            if (err.empty())
                err = "tensor '" + n + "' has wrong dtype/size (" + std::to_string(it->second.numel()) + " vs " +
                      std::to_string(numel) + ")";
            return nullptr;
        }
        return &it->second;
    }
    std::vector<float> vec(const std::string& n, size_t numel) {
        auto t = get(n, numel);
        return t ? std::vector<float>(t->f32(), t->f32() + numel) : std::vector<float>(numel, 0.f);
    }
    float* upload(const std::vector<float>& v) {
        void* d = nullptr;
        if (hipMalloc(&d, v.size() * sizeof(float) + 64) != hipSuccess) {
            if (err.empty()) err = "hipMalloc failed (weights)";
            return nullptr;
        }
        allocs.push_back(d);
        if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            if (err.empty()) err = "hipMemcpy failed (weights)";
        return static_cast<float*>(d);
    }
    void* dalloc(size_t bytes) {
        void* d = nullptr;
        if (hipMalloc(&d, bytes + 256) != hipSuccess) {
            if (err.empty()) err = "hipMalloc failed (workspace, " + std::to_string(bytes >> 20) + " MiB)";
            return nullptr;
        }
        allocs.push_back(d);
        return d;
    }
    float* falloc(size_t floats) { return static_cast<float*>(dalloc(floats * 4)); }
};

// Pack GEMM rows into the per-lane A-fragment order conv_mfma_kernel streams:
//   [wave tile (64 rows)][chunk][k-group (4 k-steps)][M-subtile (2)][lane (64)][4 floats]
//   lane l holds row (l&31) of its M-subtile, channel parity (l>>5); k-step s = tap*(CK/2) + channel pair.
template <class F>
std::vector<float> pack_A(int rows, int Cin, int KS, int CK, F get) {
    int mt64 = (rows + 63) / 64;
    mt64 = (mt64 + 1) & ~1;   // even number of wave tiles so WM=2 blocks never read past the end
    const int nch = (Cin + CK - 1) / CK;
    const int ksteps = KS * KS * CK / 2, ng = ksteps / 4, half = CK / 2;
    std::vector<float> dst((size_t)mt64 * nch * ng * 2 * 64 * 4, 0.f);
    size_t o = 0;
    for (int mt = 0; mt < mt64; ++mt)
        for (int ch = 0; ch < nch; ++ch)
            for (int g = 0; g < ng; ++g)
                for (int ms = 0; ms < 2; ++ms)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int q = 0; q < 4; ++q, ++o) {
                            const int s = g * 4 + q, t = s / half, cp = s % half;
                            const int row = mt * 64 + ms * 32 + (lane & 31);
                            const int ci = ch * CK + 2 * cp + (lane >> 5);
                            if (row < rows && ci < Cin) dst[o] = get(row, ci, t);
                        }
    return dst;
}

// f16x3 split-operand packing for conv_sh16_kernel:
//   [wave tile 64 rows][chunk 16 ch][tap][M-subtile][hi|lo][lane][8 halfs]; lane l: row (l&31), channels 8*(l>>5)+e
// Every row is scaled by its own power of two 2^k[row] before the hi/lo split so that max|row| * 2^k lies in [2^14, 2^15)
// (sh16.h: an element 2^18 times smaller than the row maximum still keeps 22 significand bits); `wscale[row]` receives the
// exact inverse 2^-k[row], which the consuming epilogue multiplies the accumulator by.  `kmax` (optional, per row) caps
// the exponent (two operands sharing an accumulator must share 2^k * s_in: see sh16_row_exponents).
inline int sh16_row_exponent(float rowmax) {
    if (!(rowmax > 0.f) || !std::isfinite(rowmax)) return 0;
    int e;
    (void)std::frexp(rowmax, &e);          // rowmax = m * 2^e, m in [0.5, 1)
    return 15 - e;                         // rowmax * 2^k in [2^14, 2^15)
}
template <class F>
std::vector<int> sh16_row_exponents(int rows, int Cin, int KS, F get) {
    std::vector<int> k(rows);
    for (int r = 0; r < rows; ++r) {
        float mx = 0.f;
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < KS * KS; ++t) mx = std::max(mx, std::fabs(get(r, ci, t)));
        k[r] = sh16_row_exponent(mx);
    }
    return k;
}
template <class F>
std::vector<float> pack_A_sh16(int rows, int Cin, int KS, F get, const std::vector<int>& kexp, bool bf16 = false) {
    const int mt64 = (rows + 63) / 64, nch = (Cin + 15) / 16, nt = KS * KS;
    std::vector<_Float16> dst((size_t)mt64 * nch * nt * 2 * 2 * 64 * 8, (_Float16)0.f);
    size_t o = 0;
    for (int mt = 0; mt < mt64; ++mt)
        for (int ch = 0; ch < nch; ++ch)
            for (int t = 0; t < nt; ++t)
                for (int ms = 0; ms < 2; ++ms) {
                    for (int hl = 0; hl < 2; ++hl)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e, ++o) {
                                const int row = mt * 64 + ms * 32 + (lane & 31);
                                const int ci = ch * 16 + (lane >> 5) * 8 + e;
                                if (row < rows && ci < Cin) {
                                    const float w = std::ldexp(get(row, ci, t), kexp[row]);
                                    if (bf16) {            // bf16 single-term mode: hi = bf16 bits (round to nearest even), lo unused
                                        uint32_t u;
                                        std::memcpy(&u, &w, 4);
                                        const uint16_t b = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                                        _Float16 hb;
                                        std::memcpy(&hb, &b, 2);
                                        dst[o] = hl == 0 ? hb : (_Float16)0.f;
                                        continue;
                                    }
                                    const _Float16 h = (_Float16)w;
                                    dst[o] = hl == 0 ? h : (_Float16)(w - (float)h);
                                }
                            }
                }
    std::vector<float> out(dst.size() / 2);
    std::memcpy(out.data(), dst.data(), dst.size() * 2);
    return out;
}
// inverse row scales 2^-k, padded to whole 64-row tiles (epilogues read float4 runs)
inline std::vector<float> sh16_wscale(const std::vector<int>& kexp) {
    std::vector<float> v(((kexp.size() + 63) / 64) * 64, 1.f);
    for (size_t r = 0; r < kexp.size(); ++r) v[r] = std::ldexp(1.f, -kexp[r]);
    return v;
}

// ---- generic conv layer on the MFMA kernel ------------------------------------------------------------------
struct ConvLayer {
    float* wpk = nullptr;
    float* bias = nullptr;
    int Cout = 0, Cin = 0, KS = 0, stride = 1, pad = 0;
    float* sh_wpk = nullptr;       // f16x3 packing (make_conv_sh16 / make_conv_s2d): split-operand A fragments + per-row inverse scales
    float* sh_wscale = nullptr;
    int s2d_cr = 0, s2d_phase0 = 0; // make_conv_s2d: real (padded) input channels and first phase of the space-to-depth form
    float* wino = nullptr;          // 3x3 stride-1 pad-1 layers of the exact-f32 path: Winograd F(2x2,3x3) A images (conv_wino.h) ...
    float* zero = nullptr;          // ... and a few zero words (source of out-of-image patch elements)
};

// w: [Cout][Cin][KS][KS] (already folded: BN / spectral norm / flips), bias may be empty
// want_wino: also pack the Winograd F(2x2,3x3) image (16 / 9 of the weights, packed through a transient host copy in double) -- only
// for layers that can reach a Winograd-sized output (>= 16 x 16) on a handle with aux.wino on (the 4 x 4 layer of the
// shape decoder, 2048 -> 2048, carried 268 MB that no launch ever read)
inline ConvLayer make_conv(Builder& B, const std::vector<float>& w, const std::vector<float>& bias, int cout, int cin,
                           int ks, int stride, int pad, bool want_wino = true) {
    ConvLayer L;
    L.Cout = cout;
    L.Cin = cin;
    L.KS = ks;
    L.stride = stride;
    L.pad = pad;
    const int ck = conv_ck(ks, stride);
    const float* wp = w.data();
    L.wpk = B.upload(pack_A(cout, cin, ks, ck, [&](int row, int ci, int t) {
        return wp[((size_t)row * cin + ci) * ks * ks + t];
    }));
    if (!bias.empty()) L.bias = B.upload(bias);
    if (want_wino && ks == 3 && stride == 1 && pad == 1 && cin % 8 == 0 && cout >= 16) {
        // the same conv as Winograd F(2x2,3x3) on the f32 matrix cores (run_conv takes it where the shape fits the kernel's tiles)
        L.wino = B.upload(pack_wino_A(cout, cin, [&](int row, int ci, int t) { return wp[((size_t)row * cin + ci) * 9 + t]; }));
        L.zero = B.upload(std::vector<float>(64, 0.f));
    }
    return L;
}

// stride-1 conv for the f16x3 kernels (conv_sh16.h): rows padded to a multiple of 4 with zero rows (C4 output runs)
inline ConvLayer make_conv_sh16(Builder& B, const std::vector<float>& w, const std::vector<float>& bias, int cout, int cin,
                                int ks, int pad) {
    ConvLayer L;
    L.Cout = (cout + 3) & ~3;
    L.Cin = cin;
    L.KS = ks;
    L.stride = 1;
    L.pad = pad;
    const float* wp = w.data();
    auto getw = [&](int row, int ci, int t) { return row < cout ? wp[((size_t)row * cin + ci) * ks * ks + t] : 0.f; };
    const auto kexp = sh16_row_exponents(L.Cout, cin, ks, getw);
    L.sh_wpk = B.upload(pack_A_sh16(L.Cout, cin, ks, getw, kexp));
    L.sh_wscale = B.upload(sh16_wscale(kexp));
    if (!bias.empty()) {
        std::vector<float> b(L.Cout, 0.f);
        std::copy(bias.begin(), bias.end(), b.begin());
        L.bias = B.upload(b);
    }
    return L;
}

// stride-2, pad-(ks > 1) conv in the space-to-depth form of conv_sh16_kernel<S2D> (conv_sh16.h): a 2x2-tap (ks 3 / 4) or
// 1x1 (ks 1) stride-1 conv at the output resolution over `phases` copies of the input channels.  Virtual channel
// vc = phase * cr + c; tap (dy, dx) of phase (py, px) carries the original weight at (ky, kx) = (2 dy + py, 2 dx + px)
// (zero where that index leaves the ks x ks kernel).  cr = cin rounded up to a multiple of 16 (padding channels: zeros).
// L.Cin = phases * cr, L.KS = 2 or 1, L.stride = 2 (marker), L.Cout padded to a multiple of 4.
inline ConvLayer make_conv_s2d(Builder& B, const std::vector<float>& w, const std::vector<float>& bias, int cout, int cin, int ks) {
    ConvLayer L;
    const int cr = (cin + 15) & ~15, phases = ks == 1 ? 1 : 4;
    L.Cout = (cout + 3) & ~3;
    L.Cin = phases * cr;
    L.KS = ks == 1 ? 1 : 2;
    L.stride = 2;
    L.pad = ks == 1 ? 0 : 1;
    L.s2d_cr = cr;
    L.s2d_phase0 = ks == 1 ? 3 : 0;
    const float* wp = w.data();
    auto getw = [&](int row, int vc, int t) {
        const int c = vc % cr;
        if (row >= cout || c >= cin) return 0.f;
        if (ks == 1) return wp[(size_t)row * cin + c];
        const int ph = vc / cr, ky = 2 * (t / 2) + (ph >> 1), kx = 2 * (t % 2) + (ph & 1);
        return (ky < ks && kx < ks) ? wp[(((size_t)row * cin + c) * ks + ky) * ks + kx] : 0.f;
    };
    const auto kexp = sh16_row_exponents(L.Cout, L.Cin, L.KS, getw);
    L.sh_wpk = B.upload(pack_A_sh16(L.Cout, L.Cin, L.KS, getw, kexp));
    L.sh_wscale = B.upload(sh16_wscale(kexp));
    if (!bias.empty()) {
        std::vector<float> b(L.Cout, 0.f);
        std::copy(bias.begin(), bias.end(), b.begin());
        L.bias = B.upload(b);
    }
    return L;
}

struct ConvOpts {
    int pad_mode = PAD_ZERO;
    int in_mode = IN_DIRECT;
    int act = ACT_NONE;
    const float* res = nullptr;
    int res_up = 0;
    int res_after_act = 0;
    float* partial = nullptr;      // split-K scratch (optional): enables split-K for few-tile / long-K layers
    long long partial_cap = 0;
    int no_wino = 0;               // 1 = keep a 3x3 stride-1 layer on the direct kernel (A/B tests)
};

inline int conv_out_size(const ConvLayer& L, int in, int in_mode) {
    const int l = in_mode == IN_DIRECT ? in : 2 * in;
    return (l + 2 * L.pad - L.KS) / L.stride + 1;
}

// in: [B][Cin][Hin][Win] -> out: [B][Cout][Ho][Wo]
inline hipError_t run_conv(const ConvLayer& L, const float* in, float* out, int B, int Hin, int Win,
                           const ConvOpts& o, hipStream_t st) {
    {   // Winograd F(2x2,3x3) route (exact f32: the same sums in another association, conv_wino.h): 3x3 stride-1 pad-1 layers whose
        // output fits the kernel's tiles of 32 x 16 pixels (16 x 16 images: pairs of samples), direct or nearest-x2 input view
        const int H = conv_out_size(L, Hin, o.in_mode), W = conv_out_size(L, Win, o.in_mode);
        const bool up = o.in_mode == IN_UP2_NEAREST, refl = o.pad_mode == PAD_REFLECT;
        const bool tiles = wino_supported(H, W, L.Cin) || (!up && !refl && !o.res_up && wino_supported_pair16(B, H, W, L.Cin));
        if (L.wino && L.zero && !o.no_wino && L.KS == 3 && L.stride == 1 && L.pad == 1 && (o.in_mode == IN_DIRECT || up) && !(up && refl) &&
            !o.res_after_act && tiles) {
            WinoParams q{};
            q.in = in;
            q.wpk = L.wino;
            q.out = out;
            q.B = B;
            q.Cin = L.Cin;
            q.Cout = L.Cout;
            q.H = H;
            q.W = W;
            q.bias = L.bias;
            q.res = o.res;
            q.res_up = o.res_up;
            q.act = o.act;
            q.reflect = refl ? 1 : 0;
            q.in_up = up ? 1 : 0;
            q.zero = L.zero;
            q.partial = o.partial;             // (few tasks, long k-loop: conv_wino_plain splits K)
            q.partial_cap = o.partial_cap;
            return conv_wino_plain(q, st);
        }
    }
    ConvParams p{};
    p.in = in;
    p.wpk = L.wpk;
    p.out = out;
    p.B = B;
    p.Cin = L.Cin;
    p.Hin = Hin;
    p.Win = Win;
    p.H = conv_out_size(L, Hin, o.in_mode);
    p.W = conv_out_size(L, Win, o.in_mode);
    p.pad = L.pad;
    p.pad_mode = o.pad_mode;
    p.in_mode = o.in_mode;
    p.Mrows = L.Cout;
    p.bias = L.bias;
    p.res = o.res;
    p.res_up = o.res_up;
    p.res_after_act = o.res_after_act;
    p.act = o.act;
    p.partial = o.partial;
    p.partial_cap = o.partial_cap;
    if (L.stride == 2) return conv_plain_s2(p, L.KS, st);
    if (L.KS == 3) return conv_plain3(p, st);
    if (L.KS == 1) return conv_plain1(p, st);
    return hipErrorInvalidValue;
}

}  // namespace chk
