// sean_model.cpp -- weight folding/packing + forward schedule of the SEAN generator on MI355X.
//
// Reference behaviour restated (file:line refer to /root/reference):
//   generator.py:72-109            SPADEGenerator.forward (fc, 7 ResBlocks, nearest x2 ups, conv_img, tanh)
//   architecture.py:69-96          SPADEResnetBlock.forward / shortcut / actvn
//   normalization.py:108-189       ACE.forward (noise, eval-BN, style broadcast, style convs, SPADE, blend, modulate)
//   normalization.py:249-257       SPADE.forward
//   torch spectral_norm (eval)     W = W_orig / (u . W_mat v)
// Exact reformulations used: convs whose input is a one-hot label map or the piecewise-constant
// style map are evaluated as 9-tap label-table gathers; sigmoid(blending) is folded into weights/biases.
#include "sean_model.h"

#include <cmath>
#include <cstring>

#include "ace_sparse.h"
#include "conv_ace_sparse.h"
#include "conv_mfma.h"
#include "conv_sh16.h"
#include "conv_pw.h"
#include "conv_wino.h"
#include "conv_wino4v.h"
#include "kernels.h"
#include "sh16.h"

namespace chk {

namespace {

constexpr int LABEL_NC = 19, STYLE = 512, HID = 128;

}  // namespace

size_t SeanModel::noise_floats(int S) const {
    size_t n = 0;
    for (const auto& b : blocks) {
        const size_t r = (size_t)S / b.res_div;
        n += r * r * (b.learned ? 3 : 2);
    }
    return n;
}

std::string SeanModel::build(const TensorStore& ts, int mb, int ms) {
    Builder B(ts, allocs);
    auto itfc = ts.find("fc.weight");
    if (itfc == ts.end() || itfc->second.shape.size() != 4) return "missing tensor 'fc.weight'";
    ngf = (int)itfc->second.shape[0] / 16;
    if (ngf < 4 || (ngf & 3)) return "unsupported ngf";
    if (ms % 32 != 0 || ms < 32 || mb < 1) return "max_size must be a multiple of 32 and max_batch >= 1";
    max_batch = mb;
    max_size = ms;

    // ---- fc as label table: T[(j*9+t)*K + k] = W[k][j][t] --------------------------------------------------
    {
        const int K = 16 * ngf;
        auto w = B.get("fc.weight", (size_t)K * LABEL_NC * 9);
        auto b = B.get("fc.bias", K);
        if (!w || !b) return B.err;
        std::vector<float> T((size_t)LABEL_NC * 9 * K);
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < LABEL_NC; ++j)
                for (int t = 0; t < 9; ++t) T[((size_t)j * 9 + t) * K + k] = w->f32()[((size_t)k * LABEL_NC + j) * 9 + t];
        fc_table = B.upload(T);
        fc_bias = B.upload(std::vector<float>(b->f32(), b->f32() + K));
    }
    {
        auto w = B.get("conv_img.weight", (size_t)3 * ngf * 9);
        auto b = B.get("conv_img.bias", 3);
        if (!w || !b) return B.err;
        img_w = B.upload(std::vector<float>(w->f32(), w->f32() + 3 * ngf * 9));
        {
            std::vector<float> w4((size_t)(ngf / 4) * 27 * 4);
            for (int g = 0; g < ngf / 4; ++g)
                for (int t = 0; t < 9; ++t)
                    for (int co = 0; co < 3; ++co)
                        for (int e = 0; e < 4; ++e)
                            w4[(((size_t)g * 9 + t) * 3 + co) * 4 + e] = w->f32()[((size_t)co * ngf + g * 4 + e) * 9 + t];
            img_w4 = B.upload(w4);
        }
        img_b = B.upload(std::vector<float>(b->f32(), b->f32() + 3));
    }

    struct BS { const char* name; int fin, fout, res_div; bool up, styled; };
    const BS specs[7] = {{"head_0", 16, 16, 32, false, true}, {"G_middle_0", 16, 16, 16, true, true},
                         {"G_middle_1", 16, 16, 16, false, true}, {"up_0", 16, 8, 8, true, true},
                         {"up_1", 8, 4, 4, true, true}, {"up_2", 4, 2, 2, true, true},
                         {"up_3", 2, 1, 1, true, false}};
    int ace_index = 0;
    blocks.clear();
    for (const auto& s : specs) {
        BlockW bw;
        bw.name = s.name;
        bw.fin = s.fin * ngf;
        bw.fout = s.fout * ngf;
        bw.fmid = bw.fin < bw.fout ? bw.fin : bw.fout;
        bw.res_div = s.res_div;
        bw.up_before = s.up;
        bw.styled = s.styled;
        bw.learned = bw.fin != bw.fout;

        // spectral norm folded at load (torch eval semantics W / (u . W_mat v)); f16x3 path: per-row power-of-two scaling
        struct SN { const float* w = nullptr; double sigma = 1.0; int cout = 0, cin = 0, ks = 0; };
        auto sn_read = [&](const std::string& p, int cout, int cin, int ks) {
            SN r;
            const size_t kk = (size_t)cin * ks * ks;
            auto w = B.get(p + ".weight_orig", (size_t)cout * kk);
            auto u = B.get(p + ".weight_u", cout);
            auto v = B.get(p + ".weight_v", kk);
            if (!w || !u || !v) return r;
            double sigma = 0.0;   // u . (W_mat v)
            for (int o = 0; o < cout; ++o) {
                double acc = 0.0;
                const float* wr = w->f32() + (size_t)o * kk;
                for (size_t i = 0; i < kk; ++i) acc += (double)wr[i] * v->f32()[i];
                sigma += acc * u->f32()[o];
            }
            r.w = w->f32();
            r.sigma = sigma;
            r.cout = cout; r.cin = cin; r.ks = ks;
            return r;
        };
        auto sn_get = [](const SN& r) {
            return [r](int row, int ci, int t) { return r.w[((size_t)row * r.cin + ci) * r.ks * r.ks + t] / (float)r.sigma; };
        };
        auto sn_pack = [&](const SN& r, const std::string& p, bool bias, const std::vector<int>* kexp, ConvW& cw) {
            if (!r.w) return;
            auto getw = sn_get(r);
            if (use_sh16) {
                cw.wpk = B.upload(pack_A_sh16(r.cout, r.cin, r.ks, getw, *kexp, terms == 2));
                cw.wscale = B.upload(sh16_wscale(*kexp));
            } else {
                cw.wpk = B.upload(pack_A(r.cout, r.cin, r.ks, r.ks == 3 ? CK_KS3 : CK_KS1, getw));
                if (wino && r.ks == 3 && r.cin % 8 == 0) cw.wino = B.upload(pack_wino_A(r.cout, r.cin, getw));
                if (wino >= 2 && r.ks == 3 && r.cin % 8 == 0 && r.cin >= 16) cw.wino4 = B.upload(pack_wino4_A(r.cout, r.cin, getw));
                if (wino && r.ks == 1 && r.cin % 16 == 0) cw.pw = B.upload(pack_pw_A(r.cout, r.cin, [&](int row, int ci) { return getw(row, ci, 0); }));
            }
            cw.Cout = r.cout;
            cw.Cin = r.cin;
            cw.KS = r.ks;
            if (bias) {
                auto bb = B.get(p + ".bias", r.cout);
                if (bb) cw.bias = B.upload(std::vector<float>(bb->f32(), bb->f32() + r.cout));
            }
        };
        int fuse_shift = 0;
        {
            const SN c0 = sn_read(bw.name + ".conv_0", bw.fmid, bw.fin, 3), c1 = sn_read(bw.name + ".conv_1", bw.fout, bw.fmid, 3);
            SN cs;
            if (bw.learned) cs = sn_read(bw.name + ".conv_s", bw.fout, bw.fin, 1);
            if (!B.err.empty()) return B.err;
            std::vector<int> k0, k1, ks_;
            if (use_sh16) {
                k0 = sh16_row_exponents(c0.cout, c0.cin, 3, sn_get(c0));
                k1 = sh16_row_exponents(c1.cout, c1.cin, 3, sn_get(c1));
                if (bw.learned) {
                    // conv_s is folded into conv_1 as extra K chunks on the same accumulators: row by row the two operands
                    // must share 2^k * s_in.  conv_1's rows keep their own exponents; conv_s's rows take k1[row] - D with
                    // D = max over rows of (k1 - ks) (no row overflows, rows of similar relative size lose nothing), and the
                    // 2^D is carried by the first-pass scale of the shortcut activations (ace_s writes hs * 8 * 2^D).
                    ks_ = sh16_row_exponents(cs.cout, cs.cin, 1, sn_get(cs));
                    int D = k1[0] - ks_[0];
                    for (int r = 0; r < c1.cout; ++r) D = std::max(D, k1[r] - ks_[r]);
                    D = std::max(-24, std::min(24, D));
                    for (int r = 0; r < c1.cout; ++r) {       // exactly ks = k1 - D, neither above its own optimum
                        ks_[r] = std::min(ks_[r], k1[r] - D);
                        k1[r] = ks_[r] + D;
                    }
                    fuse_shift = D;
                }
            }
            sn_pack(c0, bw.name + ".conv_0", true, &k0, bw.conv_0);
            sn_pack(c1, bw.name + ".conv_1", true, &k1, bw.conv_1);
            if (bw.learned) sn_pack(cs, bw.name + ".conv_s", false, &ks_, bw.conv_s);
        }
        if (!B.err.empty()) return B.err;

        auto ace = [&](const std::string& p, int C, AceW& a, float out_scale = SH16_ACT_SCALE) {
            a.out_scale = out_scale;
            a.name = p;
            a.C = C;
            a.res_div = s.res_div;
            a.styled = s.styled;
            a.index = ace_index++;
            auto nvar = B.get(p + ".noise_var", C);
            auto rm = B.get(p + ".param_free_norm.running_mean", C);
            auto rv = B.get(p + ".param_free_norm.running_var", C);
            auto ws = B.get(p + ".Spade.mlp_shared.0.weight", (size_t)HID * LABEL_NC * 9);
            auto bs = B.get(p + ".Spade.mlp_shared.0.bias", HID);
            auto wg = B.get(p + ".Spade.mlp_gamma.weight", (size_t)C * HID * 9);
            auto bg = B.get(p + ".Spade.mlp_gamma.bias", C);
            auto wb = B.get(p + ".Spade.mlp_beta.weight", (size_t)C * HID * 9);
            auto bb = B.get(p + ".Spade.mlp_beta.bias", C);
            auto blg = B.get(p + ".blending_gamma", 1);
            auto blb = B.get(p + ".blending_beta", 1);
            if (!nvar || !rm || !rv || !ws || !bs || !wg || !bg || !wb || !bb || !blg || !blb) return;
            float ag = 0.f, ab = 0.f;   // weight of the style branch
            if (a.styled) {
                ag = 1.f / (1.f + std::exp(-blg->f32()[0]));
                ab = 1.f / (1.f + std::exp(-blb->f32()[0]));
            }
            // eval BN (eps 1e-5) + noise scaling as per-channel affine
            std::vector<float> va(C), vd(C), vn(C), vbg(C), vbb(C);
            for (int c = 0; c < C; ++c) {
                const float rstd = 1.f / std::sqrt(rv->f32()[c] + 1e-5f);
                va[c] = rstd;
                vd[c] = -rm->f32()[c] * rstd;
                vn[c] = nvar->f32()[c] * rstd;
                vbg[c] = (1.f - ag) * bg->f32()[c];
                vbb[c] = (1.f - ab) * bb->f32()[c];
            }
            // mlp_shared as label table
            std::vector<float> T((size_t)LABEL_NC * 9 * HID);
            for (int k = 0; k < HID; ++k)
                for (int j = 0; j < LABEL_NC; ++j)
                    for (int t = 0; t < 9; ++t)
                        T[((size_t)j * 9 + t) * HID + k] = ws->f32()[((size_t)k * LABEL_NC + j) * 9 + t];
            a.actv_table = B.upload(T);
            a.actv_bias = B.upload(std::vector<float>(bs->f32(), bs->f32() + HID));
            {   // relu(bias + sum of one table entry per tap) <= bias + sum_t max(0, max_j T[j][t][k]): a bound that holds for
                // every label map, so the SH16 scale of the hidden activations can never saturate
                float bound = 0.f;
                for (int k = 0; k < HID; ++k) {
                    float v = bs->f32()[k];
                    for (int t = 0; t < 9; ++t) {
                        float mx = 0.f;
                        for (int j = 0; j < LABEL_NC; ++j) mx = std::max(mx, T[((size_t)j * 9 + t) * HID + k]);
                        v += mx;
                    }
                    bound = std::max(bound, v);
                }
                a.actv_scale = sh16_scale_for_bound(bound);
            }
            // SPADE gamma/beta rows: 64-row tiles = (gamma of 32 channels | beta of the same 32 channels)
            const int tiles = (C + 31) / 32;
            const float* wgp = wg->f32();
            const float* wbp = wb->f32();
            const float sg = 1.f - ag, sb = 1.f - ab;
            auto getsp = [&](int row, int ci, int t) {
                const int c = (row / 64) * 32 + (row & 31);
                if (c >= C) return 0.f;
                const bool beta = (row & 32) != 0;
                const float w = (beta ? wbp : wgp)[((size_t)c * HID + ci) * 9 + t];
                return w * (beta ? sb : sg);
            };
            std::vector<double> edge_hv, edge_W6;      // straight-edge tables: host operands (filled below, contracted after the biases are final)
            {   // per-label constants of the SPADE gamma/beta for pixels with a uniform 5x5 label neighbourhood (ace_sparse.h):
                // every tap sees a_j = relu(b_shared + sum_t' W_shared[:, j, t']), so gamma_j = (sum_t W[:, :, t]) a_j.  Double.
                std::vector<double> aj((size_t)LABEL_NC * HID);
                for (int j = 0; j < LABEL_NC; ++j)
                    for (int k = 0; k < HID; ++k) {
                        double v = bs->f32()[k];
                        for (int t = 0; t < 9; ++t) v += T[((size_t)j * 9 + t) * HID + k];
                        aj[(size_t)j * HID + k] = v > 0.0 ? v : 0.0;
                    }
                std::vector<float> gc((size_t)LABEL_NC * 2 * C);
                std::vector<double> wsum(HID);
                for (int gb = 0; gb < 2; ++gb)
                    for (int c = 0; c < C; ++c) {
                        const float* w = (gb ? wbp : wgp) + (size_t)c * HID * 9;
                        for (int k = 0; k < HID; ++k) {
                            double acc = 0.0;
                            for (int t = 0; t < 9; ++t) acc += w[k * 9 + t];
                            wsum[k] = acc;
                        }
                        for (int j = 0; j < LABEL_NC; ++j) {
                            double acc = 0.0;
                            for (int k = 0; k < HID; ++k) acc += wsum[k] * aj[(size_t)j * HID + k];
                            gc[((size_t)j * 2 + gb) * C + c] = (float)(acc * (gb ? sb : sg));
                        }
                    }
                a.gconst = B.upload(gc);
                // straight-edge pixels (ace_sparse.h, option sean.edge): the per-code rows E[2888][gamma|beta][C] of the ACEs that can run at
                // 128 pixels and more (res_div <= 4), from the hidden vectors of the windows AAB / ABB (double, host) and the column / row sums
                // of the gamma / beta weights, contracted on the device in double (ace_edge_table).  vbg / vbb (blended biases) are added.
                a.edge_tab = nullptr;
                if (edge && (use_sh16 ? sparse : wino) && s.res_div <= 4) {
                    edge_hv.assign((size_t)2 * 741 * HID, 0.0);
                    edge_W6.assign((size_t)2 * HID * 6 * C, 0.0);
                    std::vector<double>&hv = edge_hv, &W6 = edge_W6;
                    for (int o = 0; o < 2; ++o) {
                        auto line = [&](int j, int u, int k) {       // shared conv's taps of label j summed across the split: column u (o = 0) / row u (o = 1)
                            double v = 0.0;
                            for (int w = 0; w < 3; ++w) v += T[((size_t)j * 9 + (o == 0 ? w * 3 + u : u * 3 + w)) * HID + k];
                            return v;
                        };
                        for (int j = 0; j < LABEL_NC; ++j)
                            for (int k = 0; k < HID; ++k) hv[((size_t)o * 741 + j) * HID + k] = aj[(size_t)j * HID + k];
                        for (int A = 0; A < LABEL_NC; ++A)
                            for (int Bl = 0; Bl < LABEL_NC; ++Bl) {
                                if (A == Bl) continue;
                                for (int k = 0; k < HID; ++k) {
                                    const double aab = bs->f32()[k] + line(A, 0, k) + line(A, 1, k) + line(Bl, 2, k);
                                    const double abb = bs->f32()[k] + line(A, 0, k) + line(Bl, 1, k) + line(Bl, 2, k);
                                    hv[((size_t)o * 741 + 19 + (A * 19 + Bl) * 2 + 0) * HID + k] = aab > 0.0 ? aab : 0.0;
                                    hv[((size_t)o * 741 + 19 + (A * 19 + Bl) * 2 + 1) * HID + k] = abb > 0.0 ? abb : 0.0;
                                }
                            }
                    }
                    for (int gb = 0; gb < 2; ++gb)
                        for (int c = 0; c < C; ++c) {
                            const float* w = (gb ? wbp : wgp) + (size_t)c * HID * 9;
                            for (int k = 0; k < HID; ++k)
                                for (int d = 0; d < 3; ++d) {
                                    double col = 0.0, row = 0.0;
                                    for (int u = 0; u < 3; ++u) {
                                        col += w[k * 9 + u * 3 + d];      // taps (dy = u - 1, dx = d - 1)
                                        row += w[k * 9 + d * 3 + u];      // taps (dy = d - 1, dx = u - 1)
                                    }
                                    W6[(((size_t)gb * HID + k) * 6 + d) * C + c] = col;
                                    W6[(((size_t)gb * HID + k) * 6 + 3 + d) * C + c] = row;
                                }
                        }
                }
            }
            if (use_sh16) {
                auto kexp = sh16_row_exponents(tiles * 64, HID, 3, getsp);
                for (int t0 = 0; t0 < tiles * 64; t0 += 64) {     // one power of two per 64-row wave tile (a scalar in the ACE epilogue)
                    int km = kexp[t0];
                    for (int r = 0; r < 64; ++r) km = std::min(km, kexp[t0 + r]);
                    for (int r = 0; r < 64; ++r) kexp[t0 + r] = km;
                }
                a.spade_wpk = B.upload(pack_A_sh16(tiles * 64, HID, 3, getsp, kexp, terms == 2));
                a.spade_wscale = B.upload(sh16_wscale(kexp));
            } else {
                a.spade_wpk = B.upload(pack_A(tiles * 64, HID, 3, CK_KS3, getsp));
                if (wino) {      // Winograd path: row tiles of 16 channels, rows 0-15 = gamma, 16-31 = beta (conv_wino.h)
                    auto getw = [&](int row, int ci, int t) {
                        const int c = (row / 32) * 16 + (row & 15);
                        if (c >= C) return 0.f;
                        const bool beta = (row & 16) != 0;
                        return (beta ? wbp : wgp)[((size_t)c * HID + ci) * 9 + t] * (beta ? sb : sg);
                    };
                    a.spade_wino = B.upload(pack_wino_A(((C + 15) / 16) * 32, HID, getw));
                    if (wino >= 2 && wino4_ace_max_r >= 32 && C % 2 == 0) {      // (any level can fall to <= wino4_ace_max_r pixels at a smaller image size)
                        // F(4x4,3x3) image for the low-resolution levels: rows interleaved so that a lane holds gamma and beta of two channels
                        auto getw4 = [&](int row, int ci, int t) {
                            int c, beta;
                            wino4_ace_row(row, c, beta);
                            if (c >= C) return 0.f;
                            return (beta ? wbp : wgp)[((size_t)c * HID + ci) * 9 + t] * (beta ? sb : sg);
                        };
                        a.spade_wino4 = B.upload(pack_wino4_A(((C + 15) / 16) * 32, HID, getw4));
                    }
                }
            }
            if (a.styled) {
                auto cg = B.get(p + ".conv_gamma.weight", (size_t)C * STYLE * 9);
                auto cgb = B.get(p + ".conv_gamma.bias", C);
                auto cb = B.get(p + ".conv_beta.weight", (size_t)C * STYLE * 9);
                auto cbb = B.get(p + ".conv_beta.bias", C);
                if (!cg || !cgb || !cb || !cbb) return;
                for (int c = 0; c < C; ++c) {
                    vbg[c] += ag * cgb->f32()[c];
                    vbb[c] += ab * cbb->f32()[c];
                }
                std::vector<float> fw((size_t)LABEL_NC * STYLE * STYLE), fb((size_t)LABEL_NC * STYLE);
                for (int j = 0; j < LABEL_NC; ++j) {
                    auto w = B.get(p + ".fc_mu" + std::to_string(j) + ".weight", (size_t)STYLE * STYLE);
                    auto b = B.get(p + ".fc_mu" + std::to_string(j) + ".bias", STYLE);
                    if (!w || !b) return;
                    std::memcpy(&fw[(size_t)j * STYLE * STYLE], w->f32(), sizeof(float) * STYLE * STYLE);
                    std::memcpy(&fb[(size_t)j * STYLE], b->f32(), sizeof(float) * STYLE);
                }
                a.fcmu_w = B.upload(fw);
                a.fcmu_b = B.upload(fb);
                // LUT GEMM rows (t, gamma|beta, c) <- alpha * W[c][k][t]
                const float* cgp = cg->f32();
                const float* cbp = cb->f32();
                auto getl = [&](int row, int k, int) {
                    const int t = row / (2 * C), gb = (row / C) & 1, c = row % C;
                    return gb ? ab * cbp[((size_t)c * STYLE + k) * 9 + t] : ag * cgp[((size_t)c * STYLE + k) * 9 + t];
                };
                if (use_sh16) {
                    const auto kexp = sh16_row_exponents(18 * C, STYLE, 1, getl);
                    a.lut_wpk = B.upload(pack_A_sh16(18 * C, STYLE, 1, getl, kexp, terms == 2));
                    a.lut_wscale = B.upload(sh16_wscale(kexp));
                } else {
                    a.lut_wpk = B.upload(pack_A(18 * C, STYLE, 1, CK_KS1, getl));
                    if (max_batch * LABEL_NC > 64) {      // grouped LUT build (conv_pw.h): [k][row] image
                        std::vector<float> wt((size_t)STYLE * 18 * C);
                        for (int row = 0; row < 18 * C; ++row)
                            for (int k = 0; k < STYLE; ++k) wt[(size_t)k * 18 * C + row] = getl(row, k, 0);
                        a.lut_wt = B.upload(wt);
                    }
                }
                if (max_batch * (LABEL_NC + 1) <= 64) {   // small batches (interactive use): the LUT build is a weight-streaming GEMV
                    std::vector<float> rows((size_t)18 * C * STYLE);
                    for (int row = 0; row < 18 * C; ++row) {
                        const int t = row / (2 * C), gb = (row / C) & 1, cc = row % C;
                        const float* src = gb ? cbp : cgp;
                        const float al = gb ? ab : ag;
                        // f16x3 path: the ACE epilogue expects the style LUT pre-multiplied by the output scale (conv_sh16.h)
                        const float als = use_sh16 ? al * a.out_scale : al;
                        for (int k = 0; k < STYLE; ++k) rows[(size_t)row * STYLE + k] = als * src[((size_t)cc * STYLE + k) * 9 + t];
                    }
                    a.lut_rows = B.upload(rows);
                }
            }
            if (!edge_hv.empty()) {      // (after the style branch's biases went into vbg / vbb: the table rows are complete gamma / beta)
                const std::vector<double>&hv = edge_hv, &W6 = edge_W6;
                double *dW = nullptr, *dH = nullptr;
                float *dbg_ = B.upload(vbg), *dbb_ = B.upload(vbb);
                a.edge_tab = B.falloc((size_t)ACE_EDGE_CODES * 2 * C);
                if (B.err.empty() && hipMalloc(&dW, W6.size() * 8) == hipSuccess && hipMalloc(&dH, hv.size() * 8) == hipSuccess &&
                    hipMemcpy(dW, W6.data(), W6.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
                    hipMemcpy(dH, hv.data(), hv.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
                    ace_edge_table(dW, dH, dbg_, dbb_, sg, sb, a.edge_tab, C, nullptr) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
                } else if (B.err.empty()) {
                    B.err = "edge table of " + p + " failed";
                }
                if (dW) (void)hipFree(dW);
                if (dH) (void)hipFree(dH);
            }
            a.bn_a = B.upload(va);
            a.bn_d = B.upload(vd);
            a.nv = B.upload(vn);
            a.bias_g = B.upload(vbg);
            a.bias_b = B.upload(vbb);
        };
        if (bw.learned) ace(bw.name + ".ace_s", bw.fin, bw.ace_s, std::ldexp(SH16_ACT_SCALE, fuse_shift));
        ace(bw.name + ".ace_0", bw.fin, bw.ace_0);
        ace(bw.name + ".ace_1", bw.fmid, bw.ace_1);
        if (!B.err.empty()) return B.err;
        blocks.push_back(bw);
    }

    // ---- Zencoder (optional: only when its tensors were loaded) ----------------------------------------------
    has_zencoder = false;
    if (ts.find("Zencoder.model.1.weight") != ts.end()) {
        auto plain = [&](const std::string& p, int cout, int cin, int stride, ConvLayer& L) {
            L = make_conv(B, B.vec(p + ".weight", (size_t)cout * cin * 9), B.vec(p + ".bias", cout), cout, cin, 3, stride, 1, false);      // (z14: its own reflect-padded image)
        };
        plain("Zencoder.model.1", 32, 3, 1, z1);
        z1_w = B.upload(B.vec("Zencoder.model.1.weight", (size_t)32 * 3 * 9));      // raw [32][3][3][3] for the direct stem conv
        plain("Zencoder.model.4", 64, 32, 2, z4);
        plain("Zencoder.model.7", 128, 64, 2, z7);
        {   // ConvTranspose2d(128,256,k3,s2,p1,op1): weight [in=128][out=256][3][3] -> plain conv over the
            // zero-inserted input with flipped taps: W'[co][ci][ky][kx] = Wt[ci][co][2-ky][2-kx]
            auto wt = B.vec("Zencoder.model.10.weight", (size_t)128 * 256 * 9);
            std::vector<float> w((size_t)256 * 128 * 9);
            for (int co = 0; co < 256; ++co)
                for (int ci = 0; ci < 128; ++ci)
                    for (int t = 0; t < 9; ++t) w[((size_t)co * 128 + ci) * 9 + t] = wt[((size_t)ci * 256 + co) * 9 + (8 - t)];
            z10 = make_conv(B, w, B.vec("Zencoder.model.10.bias", 256), 256, 128, 3, 1, 1, false);      // (its Winograd form is z10_wino: four phase convs)
        }
        plain("Zencoder.model.14", 512, 256, 1, z14);
        z14_wino = z10_wino = z14_wino4 = nullptr;
        if (!use_sh16 && wino) {   // exact-f32 path: the 256 -> 512 conv (91 % of the Zencoder FLOPs) as Winograd F(2x2,3x3), reflection-padded
            auto w14 = B.vec("Zencoder.model.14.weight", (size_t)512 * 256 * 9);
            const float* wp = w14.data();
            z14_wino = B.upload(pack_wino_A(512, 256, [&](int row, int ci, int t) { return wp[((size_t)row * 256 + ci) * 9 + t]; }));
            // "sean.wino" = 2: the same conv as F(4x4,3x3) (conv_wino4.h, reflection instantiation) where the half-resolution grid is a multiple of 32
            if (wino >= 2) z14_wino4 = B.upload(pack_wino4_A(512, 256, [&](int row, int ci, int t) { return wp[((size_t)row * 256 + ci) * 9 + t]; }));
            // ConvTranspose2d(128, 256, k3, s2, p1, op1) (architecture.py:167-170) as four phase convs of the INPUT grid:
            //   out[2y+py][2x+px] = sum over dy, dx in {0, 1} of x[y+dy][x+dx] * Wt[ci][co][py+1-2dy][px+1-2dx]   (taps outside 0..2 absent)
            // each a 3x3 kernel with non-zero taps at offsets 0 / +1 only; as Winograd F(2x2,3x3) that is 4 products per output pixel,
            // against 9 of the plain conv over the zero-inserted view (three quarters of them on inserted zeros).  Row = 4 co + phase.
            auto wt10 = B.vec("Zencoder.model.10.weight", (size_t)128 * 256 * 9);
            const float* w10 = wt10.data();
            z10_wino = B.upload(pack_wino_A(1024, 128, [&](int row, int ci, int t) {
                const int co = row >> 2, py = (row >> 1) & 1, px = row & 1, dy = t / 3 - 1, dx = t % 3 - 1;
                if (dy < 0 || dx < 0) return 0.f;
                const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
                return (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) ? w10[((size_t)ci * 256 + co) * 9 + ky * 3 + kx] : 0.f;
            }));
            // The same four phases as plain GEMMs (conv_pw.h) over shifted views of the input (misc_kernels.hip convt_shift4: planes
            // (0,1) | (0,0) | (1,0) | (1,1)): phase (py, px) contracts only its OWN taps -- 1 + 2 + 2 + 4 = 9 products per 2 x 2 outputs and
            // channel pair, against 16 of the Winograd phase convs above (whose 3x3 kernels are mostly zeros).  Option "sean.convt_gemm".
            if (convt_gemm) {
                static const int first[4] = {1, 0, 1, 0}, cnt[4] = {1, 2, 2, 4}, sdy[4] = {0, 0, 1, 1}, sdx[4] = {1, 0, 0, 1};
                for (int ph = 0; ph < 4; ++ph) {
                    const int py = ph >> 1, px = ph & 1;
                    z10_pw[ph] = B.upload(pack_pw_A(256, cnt[ph] * 128, [&](int co, int ci) {
                        const int sh = first[ph] + ci / 128, c = ci % 128;
                        const int ky = py + 1 - 2 * sdy[sh], kx = px + 1 - 2 * sdx[sh];
                        return (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) ? w10[((size_t)c * 256 + co) * 9 + ky * 3 + kx] : 0.f;
                    }));
                }
            }
        }
        if (use_sh16) {   // the 256->512 conv is 91 % of the Zencoder FLOPs: run it on the f16x3 path too
            auto w14 = B.vec("Zencoder.model.14.weight", (size_t)512 * 256 * 9);
            const float* wp = w14.data();
            auto g14 = [&](int row, int ci, int t) { return wp[((size_t)row * 256 + ci) * 9 + t]; };
            const auto k14 = sh16_row_exponents(512, 256, 3, g14);
            z14_sh = B.upload(pack_A_sh16(512, 256, 3, g14, k14));
            z14_ws = B.upload(sh16_wscale(k14));
            // the ConvTranspose too (a 3x3 conv over the zero-inserted view, flipped taps as above)
            auto wt10 = B.vec("Zencoder.model.10.weight", (size_t)128 * 256 * 9);
            const float* w10 = wt10.data();
            auto g10 = [&](int row, int ci, int t) { return w10[((size_t)ci * 256 + row) * 9 + (8 - t)]; };
            const auto k10 = sh16_row_exponents(256, 128, 3, g10);
            z10_sh = B.upload(pack_A_sh16(256, 128, 3, g10, k10));
            z10_ws = B.upload(sh16_wscale(k10));
            {   // depth-to-space form of the same ConvTranspose: out[2y+py][2x+px] += x[y+dy][x+dx] * Wt[ci][co][py+1-2dy][px+1-2dx]
                auto gd = [&](int row, int ci, int t) {
                    const int ph = row / 256, co = row % 256;
                    const int ky = (ph >> 1) + 1 - 2 * (t / 2), kx = (ph & 1) + 1 - 2 * (t % 2);
                    return (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) ? w10[((size_t)ci * 256 + co) * 9 + ky * 3 + kx] : 0.f;
                };
                const auto kd = sh16_row_exponents(1024, 128, 2, gd);
                z10_d2s = B.upload(pack_A_sh16(1024, 128, 2, gd, kd));
                z10_d2s_ws = B.upload(sh16_wscale(kd));
            }
            // and the two stride-2 convs (space-to-depth form)
            z4_s2d = make_conv_s2d(B, B.vec("Zencoder.model.4.weight", (size_t)64 * 32 * 9), B.vec("Zencoder.model.4.bias", 64), 64, 32, 3);
            z7_s2d = make_conv_s2d(B, B.vec("Zencoder.model.7.weight", (size_t)128 * 64 * 9), B.vec("Zencoder.model.7.bias", 128), 128, 64, 3);
        }
        if (!B.err.empty()) return B.err;
        has_zencoder = true;
    }

    amax_slots = static_cast<unsigned*>(B.dalloc(64 * sizeof(unsigned)));   // 2 per ACE: [output, style projections]
    zero_page = static_cast<float*>(B.dalloc(256));
    if (zero_page) (void)hipMemset(zero_page, 0, 256);
    splitk_cap = (long long)16 << 20;     // 64 MiB of split-K slabs (low-resolution layers only)
    splitk_ws = B.falloc((size_t)splitk_cap);
    gb_small = nullptr;
    gb_small_cap = 0;
    if (!use_sh16) {
        gb_small_cap = (long long)4 << 20;      // 16 MiB: rows x pixels of the tiny ACE launches that take the plain split-K route
        gb_small = B.falloc((size_t)gb_small_cap);
    }
    n_aces = ace_index;
    // Run-ahead mode of small jobs (Runner::prepare_all_ahead): a side stream, one join event and one set of buffers per ACE.
    // Only when the handle is sized for interactive work -- for large batches the convs own every CU and nothing co-schedules.
    // Default: interactive sizes only (up to 2 x 512^2 per chunk, where the win was measured: 3.3 -> 2.65 ms per render at
    // 256^2).  The per-ACE buffers of the full mode cost mb * r^2 * 128 * 4 bytes each -- 4.3 GB for an 8 x 512^2 handle, for
    // 1.6 % at that size -- so larger handles opt in explicitly: option "sean.ahead" = images of 512^2 per chunk (0 = never).
    if (ahead_pixels < 0) ahead_pixels = (long long)2 * 512 * 512;
    // Larger jobs keep the (HBM-write-bound) label-table kernels inline and run only the style LUT builds -- small,
    // latency-bound GEMMs -- ahead (ahead_full = false) -- unless the overlap mode serves them (sean_model.h): then the label
    // tables of every ACE run ahead too, on a side stream confined to a few CUs.
    {
        int ncu = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0) num_cus = ncu;
    }
    overlap_on = false;
    if (overlap > 0 && !use_sh16 && wino && (long long)mb * ms * ms > ahead_pixels && ms >= 128) {
        int ncu = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
        const int want = std::min(overlap, ncu / 2) & ~7;
        if (want >= 8) {
            uint32_t mask[16] = {};
            for (int i = 0; i < want; ++i) mask[i >> 5] |= 1u << (i & 31);      // (CU i of the mask sits on XCD i % 8)
            const uint32_t words = (uint32_t)((ncu + 31) / 32);
            if (hipExtStreamCreateWithCUMask(&side, words, mask) == hipSuccess && hipExtStreamCreateWithCUMask(&side_int, words, mask) == hipSuccess &&
                hipStreamCreateWithFlags(&main_i, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) == hipSuccess) {
                overlap_on = true;
            } else {
                return "overlap mode: stream creation failed (hipExtStreamCreateWithCUMask)";
            }
        }
    }
    if (overlap_on) {
        if (ahead_pixels <= 0) ahead_pixels = 1;      // (the per-ACE buffers below)
        ev_x.assign(n_aces, nullptr);
        ev_int.assign(n_aces, nullptr);
        for (auto& e : ev_x)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return "event creation failed";
        for (auto& e : ev_int)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return "event creation failed";
        claim_pool = static_cast<unsigned*>(B.dalloc((size_t)CLAIM_SLOTS * CLAIM_WORDS * sizeof(unsigned)));
    }
    if (ahead_pixels > 0) {
        ahead_full = overlap_on || (long long)mb * ms * ms <= ahead_pixels;
        if (!overlap_on && (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess))
            return "side stream creation failed";
        ev_join.assign(n_aces, nullptr);
        actv_ahead.assign(n_aces, nullptr);
        lut_ahead.assign(n_aces, nullptr);
        for (auto& e : ev_join)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return "event creation failed";
        const size_t npad_b = ((size_t)mb * (LABEL_NC + 1) + 31) / 32 * 32;
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!a) continue;
                const size_t r = (size_t)ms / a->res_div;
                if (ahead_full) actv_ahead[a->index] = B.falloc((size_t)mb * r * (r + 64) * (HID + 20));
                if (a->styled) lut_ahead[a->index] = B.falloc(npad_b * 18 * a->C);
            }
        splitk_side = B.falloc((size_t)splitk_cap);
    }

    // ---- workspace arena --------------------------------------------------------------------------------
    const size_t MB = mb, S = ms;
    for (int k = 1; k <= 5; ++k) {   // res_div 2^k
        const size_t r = S >> k;
        lab_r[k] = static_cast<uint8_t*>(B.dalloc(MB * r * r));
    }
    size_t nf = noise_floats(ms);
    noise_ws = static_cast<float*>(B.dalloc(MB * nf * 4));
    const size_t npad = ((MB * (LABEL_NC + 1) + 31) / 32) * 32;     // f16x3 path: 20 columns per sample (19 = zero column)
    mu_img = static_cast<float*>(B.dalloc((size_t)STYLE * npad * 4));
    if (mu_img) (void)hipMemset(mu_img, 0, (size_t)STYLE * npad * 4);      // pad columns stay zero
    // one fc_mu launch per chunk for all styled ACEs (f16x3: SH16 images; exact f32: f32 images, batches of more than 64 (sample, label) columns)
    fcmu_batched = use_sh16 ? max_batch * (LABEL_NC + 1) > 64 : max_batch * LABEL_NC > 64;
    if (fcmu_batched) {
        mu_stride = (long long)STYLE * npad;
        mu_all = static_cast<float*>(B.dalloc((size_t)n_aces * mu_stride * 4));
        if (mu_all) (void)hipMemset(mu_all, 0, (size_t)n_aces * mu_stride * 4);
        std::vector<const float*> wp(n_aces, nullptr), bp(n_aces, nullptr);
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1})
                if (a && a->styled) { wp[a->index] = a->fcmu_w; bp[a->index] = a->fcmu_b; }
        fcmu_w_ptrs = static_cast<const float**>(B.dalloc(n_aces * sizeof(float*)));
        fcmu_b_ptrs = static_cast<const float**>(B.dalloc(n_aces * sizeof(float*)));
        if (fcmu_w_ptrs && fcmu_b_ptrs) {
            (void)hipMemcpy(fcmu_w_ptrs, wp.data(), n_aces * sizeof(float*), hipMemcpyHostToDevice);
            (void)hipMemcpy(fcmu_b_ptrs, bp.data(), n_aces * sizeof(float*), hipMemcpyHostToDevice);
        }
    }
    lut_groups = nullptr;
    lut_ngroups = lut_group_tiles = 0;
    lut_group_rows = 0.0;
    if (fcmu_batched && !use_sh16 && lut_grouped) {
        if (lut_ahead.empty()) lut_ahead.assign(n_aces, nullptr);
        std::vector<PwGroup> g;
        bool ok = true;                           // all styled ACEs or none: the consumers take every LUT from the same place
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1})
                if (a && a->styled && (!a->lut_wt || (18 * a->C) % 128)) ok = false;
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!ok || !a || !a->styled) continue;
                if (!lut_ahead[a->index]) lut_ahead[a->index] = B.falloc(npad * 18 * a->C);
                PwGroup e{};
                e.in = a->lut_wt;
                e.wpk = mu_all + (size_t)a->index * mu_stride;
                e.out = lut_ahead[a->index];
                e.HW = 18 * a->C;                 // (C % 64 == 0: 18 C % 128 == 0)
                e.start = lut_group_tiles;        // in pixel tiles; the kernel multiplies by the call's row groups
                if (!e.out) { ok = false; continue; }
                lut_group_tiles += e.HW / 128;
                lut_group_rows += e.HW;
                g.push_back(e);
            }
        if (ok && !g.empty()) {
            lut_groups = B.dalloc(g.size() * sizeof(PwGroup));
            if (lut_groups) (void)hipMemcpy(lut_groups, g.data(), g.size() * sizeof(PwGroup), hipMemcpyHostToDevice);
            lut_ngroups = (int)g.size();
        }
    }
    size_t lutmax = 0, h0max = 0, midmax = 0, outmax = 0;
    for (const auto& b : blocks) {
        const size_t r = S / b.res_div, px = MB * r * r;
        if (b.styled) lutmax = std::max(lutmax, npad * 18 * b.fin);      // npad columns: the f16x3 LUT GEMM writes the pad too
        h0max = std::max(h0max, px * b.fin);
        midmax = std::max(midmax, px * b.fmid);
        outmax = std::max(outmax, px * b.fout);
    }
    outmax = std::max(outmax, (size_t)MB * 16 * ngf * (S / 32) * (S / 32));
    if (has_zencoder) {   // Zencoder activations live in h0/hs (<= 512 ch at S/2) and dx/h1 (<= 64 ch at S/2)
        h0max = std::max(h0max, MB * S * S * 128);
        midmax = std::max(midmax, MB * S * S * (use_sh16 ? 64 : 16));
    }
    lut = static_cast<float*>(B.dalloc(lutmax * 4));
    actv = static_cast<float*>(B.dalloc(MB * S * S * HID * 4));      // (direct kernels; the Winograd ACE levels own padded buffers, actv_lvl)
    h0 = static_cast<float*>(B.dalloc(h0max * 4));
    hs = static_cast<float*>(B.dalloc(h0max * 4));
    dx = static_cast<float*>(B.dalloc(midmax * 4));
    h1 = static_cast<float*>(B.dalloc(midmax * 4));
    xs = static_cast<float*>(B.dalloc(outmax * 4));
    xa = static_cast<float*>(B.dalloc(outmax * 4));
    xb = static_cast<float*>(B.dalloc(outmax * 4));
    // ---- Winograd ACE path: boundary-quad lists per level, task lists per (level, row tiles), per-sample style images ------
    for (int k = 0; k < 6; ++k) {
        wq_level[k] = WinoLevel();
        actv_lvl[k] = nullptr;
    }
    pad_state.clear();
    wsty = wsty4 = nullptr;
    if (wino && !use_sh16) {
        size_t wsty_max = 0, wsty4_max = 0;
        int lvl_planes[6] = {0, 0, 0, 0, 0, 0};        // planes of the level's padded hidden-activation buffer: what its ACEs write (ace_prepare: kout)
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!a || !a->spade_wino) continue;
                int k = 0;
                while ((1 << k) < a->res_div) ++k;
                const int r = ms >> k;
                if (r < 32 || r % 32) continue;
                WinoLevel& L = wq_level[k];
                if (!L.qlist) {
                    const bool gather = wino_gather && mb <= 32 && (r / 2) * (r / 2) <= 2048 * 64;
                    L.TH = gather ? 16 : wino_tile_h(r);
                    L.cap_tiles = mb * (r / 32) * (r / L.TH);
                    L.qlist = static_cast<uint8_t*>(B.dalloc((size_t)L.cap_tiles * 8 * L.TH));
                    L.qcnt = static_cast<int*>(B.dalloc((size_t)L.cap_tiles * sizeof(int)));
                    L.pcnt = static_cast<int*>(B.dalloc((size_t)L.cap_tiles * sizeof(int)));
                    if (gather) {
                        L.gq_cap = (r / 2) * (r / 2);
                        L.gq = static_cast<unsigned*>(B.dalloc((size_t)mb * L.gq_cap * sizeof(unsigned)));
                        L.gq_n = static_cast<int*>(B.dalloc(32 * sizeof(int)));
                        L.qoff = static_cast<int*>(B.dalloc((size_t)L.cap_tiles * sizeof(int)));
                        L.chunk_base = static_cast<int*>(B.dalloc(34 * sizeof(int)));
                        L.patch_mode = static_cast<int*>(B.dalloc(sizeof(int)));
                    }
                }
                const int nrt = (a->C + 15) / 16;
                bool have = false;
                for (const auto& w : L.works) have = have || w.nrt == nrt;
                if (!have) {
                    WinoWork w;
                    w.nrt = nrt;
                    w.work = static_cast<unsigned*>(B.dalloc(((size_t)L.cap_tiles * 4 + mb) * ((nrt + 1) / 2) * sizeof(unsigned)));
                    w.total = static_cast<int*>(B.dalloc(8 * sizeof(int)));
                    L.works.push_back(w);
                }
                lvl_planes[k] = std::max(lvl_planes[k], a->styled ? HID + 20 : HID);       // (up_3 is unstyled: 128 planes, no one-hot planes)
                if (a->styled) wsty_max = std::max(wsty_max, (size_t)mb * nrt * 5 * 2048);
                if (a->styled && a->spade_wino4) wsty4_max = std::max(wsty4_max, (size_t)mb * nrt * 6 * wino4::ADW);
            }
        if (wsty_max) wsty = B.falloc(wsty_max);
        // patch source of the gather kernel (conv_wino.h): up to 32 chunks of 64 boundary quads per sample (3 % of the quads of a 512^2
        // level) x 148 channels x 4 KB -- what the straight-edge reduction leaves on label maps with straight region borders
        patchbuf = nullptr;
        patch_cap_chunks = 0;
        if (patch && wino_gather && mb <= 32) {
            patch_cap_chunks = 32 * mb;
            patchbuf = B.falloc((size_t)patch_cap_chunks * (HID + 20) * 1024);
        }
        if (wsty4_max) wsty4 = B.falloc(wsty4_max);
        // pre-transformed-input route (conv_wino4v.h): one V image, sized for the largest layer that takes it at (mb, ms)
        vbuf = nullptr;
        vbuf_bytes = 0;
        if (wino >= 2 && wino4v) {
            size_t need = 0;
            for (const auto& b : blocks) {
                int k = 0;
                while ((1 << k) < b.ace_0.res_div) ++k;
                const int r = ms >> k;
                if (r < 32 || r % 32 || r > 64) continue;
                for (const ConvW* c : {&b.conv_0, &b.conv_1})
                    if (c->wino4 && wino4v_pays(c->Cout, r)) need = std::max(need, wino4v_bytes(mb, r, r, c->Cin / 4));
                for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1})
                    if (a && a->spade_wino4 && r <= wino4_ace_max_r && wino4v_pays(2 * a->C, r)) need = std::max(need, wino4v_bytes(mb, r, r, 38));
            }
            // the Zencoder's 256 -> 512 conv on the half-resolution grid: 9 bytes x 256 channels x (ms / 2)^2 per sample -- up to 2.5 GB only
            if (z14_wino4 && (ms / 2) % 32 == 0 && wino4v_bytes(mb, ms / 2, ms / 2, 64) <= ((size_t)5 << 29)) need = std::max(need, wino4v_bytes(mb, ms / 2, ms / 2, 64));
            if (need) {
                vbuf = static_cast<float*>(B.dalloc(need));
                vbuf_bytes = need;
            }
        }
        for (int k = 0; k < 6; ++k)
            if (wq_level[k].qlist) {
                const size_t r = (size_t)ms >> k;
                actv_lvl[k] = B.falloc((size_t)mb * r * wino_apitch((int)r) * lvl_planes[k]);
            }
    }
    prof_stats_cap = 16384;
    prof_stats_used = 0;
    prof_stats = static_cast<int*>(B.dalloc((size_t)prof_stats_cap * 4 * sizeof(int)));
    // ---- exact SPADE-interior reduction: classification buffers per level, work lists per (level, row tiles) ------------
    for (int k = 0; k < 6; ++k) {
        sp_level[k][0] = sp_level[k][1] = SparseLevel();
        sp_work[k].clear();
    }
    gtab = gtab_side = nullptr;
    if (sparse) {
        int cmax = 0;
        for (const auto& b : blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!a) continue;
                int k = 0;
                while ((1 << k) < a->res_div) ++k;
                const int r = ms >> k;
                if (r < sparse_min_r || r < 32) continue;
                const int mt = (a->C + 31) / 32;
                // f16x3 path: tile-skip mode of the wave-specialised kernel, whose tiles are 32 x 16
                const int th = use_sh16 ? 16 : sparse_tile_h(mt, r);
                SparseLevel& L = sp_level[k][th == 16 ? 1 : 0];
                if (!L.u5) {
                    L.TH = th;
                    L.cap_tiles = mb * ((r + 31) / 32) * ((r + L.TH - 1) / L.TH);
                    L.u5 = static_cast<uint8_t*>(B.dalloc((size_t)mb * r * r));
                    L.e16 = (edge && (use_sh16 || wino) && r >= 128) ? static_cast<uint16_t*>(B.dalloc((size_t)mb * r * r * sizeof(uint16_t))) : nullptr;
                    L.need = static_cast<uint8_t*>(B.dalloc((size_t)mb * r * r));
                    L.list = static_cast<uint16_t*>(B.dalloc((size_t)L.cap_tiles * 32 * L.TH * sizeof(uint16_t)));
                    L.cnt = static_cast<int*>(B.dalloc((size_t)L.cap_tiles * sizeof(int)));
                }
                bool have = false;
                for (const auto& w : sp_work[k]) have = have || w.mtiles == mt;
                if (!have) {
                    SparseWork w;
                    w.mtiles = mt;
                    w.TH = L.TH;
                    w.mode = use_sh16 ? (sh16_compact ? (sh16_compact >= 2 ? 2 : 3) : 1) : 0;   // 3: with pair entries
                    w.cap = (long long)L.cap_tiles * (use_sh16 ? mt : sparse_max_tasks(L.TH, mt));
                    w.work = static_cast<unsigned*>(B.dalloc((size_t)w.cap * sizeof(unsigned)));
                    w.total = static_cast<int*>(B.dalloc(4 * sizeof(int)));
                    if (w.mode == 3) {
                        w.work2 = static_cast<unsigned*>(B.dalloc((size_t)L.cap_tiles * ((mt + 1) / 2) * sizeof(unsigned)));
                        w.total2 = static_cast<int*>(B.dalloc(4 * sizeof(int)));
                        if (mt >= 4 && !(dbg & 67108864))
                            w.work3 = static_cast<unsigned*>(B.dalloc((size_t)L.cap_tiles * ((mt + 3) / 4) * sizeof(unsigned)));
                    }
                    sp_work[k].push_back(w);
                }
                cmax = std::max(cmax, a->C);
            }
        if (cmax) gtab = B.falloc((size_t)mb * LABEL_NC * 2 * cmax);
        if (cmax && overlap_on) gtab_side = B.falloc((size_t)mb * LABEL_NC * 2 * cmax);
        p6 = (cmax && edge && (use_sh16 || wino)) ? B.falloc((size_t)mb * LABEL_NC * 6 * 2 * cmax) : nullptr;      // (straight-edge pixels: ace_sparse.h)
    }
    if (!B.err.empty()) return B.err;
    if (hipDeviceSynchronize() != hipSuccess) return "hipDeviceSynchronize failed after weight upload";
    return "";
}

void SeanModel::destroy() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    for (auto e : ev_join)
        if (e) (void)hipEventDestroy(e);
    ev_join.clear();
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    ev_fork = nullptr;
    if (side) (void)hipStreamDestroy(side);
    side = nullptr;
    if (side_int) (void)hipStreamDestroy(side_int);
    if (main_i) (void)hipStreamDestroy(main_i);
    side_int = main_i = nullptr;
    for (auto e : ev_x)
        if (e) (void)hipEventDestroy(e);
    for (auto e : ev_int)
        if (e) (void)hipEventDestroy(e);
    ev_x.clear();
    ev_int.clear();
    if (ev_in) (void)hipEventDestroy(ev_in);
    if (ev_out) (void)hipEventDestroy(ev_out);
    ev_in = ev_out = nullptr;
    overlap_on = false;
    claim_pool = nullptr;
    actv_ahead.clear();
    lut_ahead.clear();
    splitk_side = nullptr;
    for (auto& r : prof) {
        ev_pool.push_back(r.e0);
        ev_pool.push_back(r.e1);
    }
    prof.clear();
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    ev_pool.clear();
    blocks.clear();
}

namespace {

struct Runner {
    SeanModel& m;
    hipStream_t st;
    std::string err;
    int B, S;
    Runner(SeanModel& mm, hipStream_t s, int b, int sz) : m(mm), st(s), B(b), S(sz) {}

    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
    }
    hipEvent_t ev() {
        if (!m.ev_pool.empty()) {
            hipEvent_t e = m.ev_pool.back();
            m.ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        check(hipEventCreate(&e), "hipEventCreate");
        return e;
    }
    template <class F>
    void timed(int kind, double flops, double bytes, F launch) { timed(kind, flops, bytes, nullptr, 0.0, 0.0, 0.0, 0.0, launch); }
    double next_flops_exec = -1.0;      // set before a timed() whose matrix cores run fewer FLOPs than the dense count
    template <class F>
    void timed(int kind, double flops, double bytes, const int* sp_stat, double sp_unit, double sp_bytes_px, double sp_bytes_fixed,
               double sp_npix, F launch) {
        if (!m.prof_on) {
            next_flops_exec = -1.0;
            launch();
            return;
        }
        ProfRec r;
        r.kind = kind;
        r.flops = flops;
        r.bytes = bytes;
        r.sp_stat = sp_stat;
        if (sp_stat && m.prof_stats && m.prof_stats_used < m.prof_stats_cap) {
            // snapshot of the work list's statistics on the stream: a later chunk / step rebuilds the list
            int* slot = m.prof_stats + 4 * (size_t)m.prof_stats_used++;
            check(hipMemcpyAsync(slot, sp_stat, 4 * sizeof(int), hipMemcpyDeviceToDevice, st), "profile statistics snapshot");
            r.sp_stat = slot;
        }
        r.sp_flops_unit = sp_unit;
        r.sp_bytes_px = sp_bytes_px;
        r.sp_bytes_fixed = sp_bytes_fixed;
        r.sp_npix = sp_npix;
        r.flops_exec = next_flops_exec;
        next_flops_exec = -1.0;
        r.e0 = ev();
        r.e1 = ev();
        check(hipEventRecord(r.e0, st), "hipEventRecord");
        launch();
        check(hipEventRecord(r.e1, st), "hipEventRecord");
        m.prof.push_back(r);
    }
    void tap_sh16(const std::string& name, const float* src, int C, size_t hw, const AceW& producer) {
        auto it = m.taps.find(name);
        if (it == m.taps.end() || !it->second) return;
        if (m.use_sh16)
            check(sh16_decode(src, it->second, B, C, (long long)hw, producer.out_scale, m.amax_slots + 2 * producer.index, st, m.terms == 2), "tap decode");
        else check(hipMemcpyAsync(it->second, src, (size_t)B * C * hw * 4, hipMemcpyDeviceToDevice, st), "tap copy");
    }
    // f32 tensors between kernels are NCHW on the exact-f32 path and C4 ([B][C/4][HW][4]) on the f16x3 path
    void tap_c4(const std::string& name, const float* src, int C, size_t hw) {
        auto it = m.taps.find(name);
        if (it == m.taps.end() || !it->second) return;
        if (m.use_sh16) check(c4_decode(src, it->second, B, C, (long long)hw, st), "tap decode");
        else check(hipMemcpyAsync(it->second, src, (size_t)B * C * hw * 4, hipMemcpyDeviceToDevice, st), "tap copy");
    }
    void tap(const std::string& name, const float* src, size_t floats) {
        auto it = m.taps.find(name);
        if (it != m.taps.end() && it->second)
            check(hipMemcpyAsync(it->second, src, floats * 4, hipMemcpyDeviceToDevice, st), "tap copy");
    }
    // overlap mode (sean_model.h): interior passes on m.side_int beside the boundary convs, dynamic task claiming in the convs
    bool overlap = false;
    int claim_next = 0;
    unsigned* next_claim() {
        if (!m.claim_pool || claim_next >= SeanModel::CLAIM_SLOTS) return nullptr;
        return m.claim_pool + (size_t)SeanModel::CLAIM_WORDS * claim_next++;
    }
    static int lk_of(const AceW& a) {
        int k = 0;
        while ((1 << k) < a.res_div) ++k;
        return k;
    }
    // exact SPADE-interior reduction: the level's classification and the work list of (level, row tiles), once per chunk
    bool lvl_done[6][2] = {};
    std::vector<int> work_done[6];
    const SparseWork* sparse_prepare(const AceW& a, const uint8_t* lab, int r) {
        int k = 0;
        while ((1 << k) < a.res_div) ++k;
        const int mt = (a.C + 31) / 32;
        if (!m.sparse || r < m.sparse_min_r || r < 32) return nullptr;
        for (const auto& w : m.sp_work[k])
            if (w.mtiles == mt) {
                const int ti = w.TH == 16 ? 1 : 0;
                const SparseLevel& L = m.sp_level[k][ti];
                if (!L.u5) return nullptr;
                const int ntiles = B * ((r + 31) / 32) * ((r + L.TH - 1) / L.TH);
                // f16x3 / f16 / bf16 path, pixel-level compaction: the interior pass knows the straight-edge pixels (ace_interior_sh16_tile_kernel)
                const bool want = m.use_sh16 && m.edge && L.e16 && r >= 128 && a.edge_tab && (!a.styled || m.p6) && w.mode >= 2;
                if (!lvl_done[k][ti] || lvl_edges[k][ti] != want) {
                    // (a level classified WITH straight-edge marks for the Winograd ACEs and then needed by the direct sparse kernels of the
                    //  exact-f32 path, which do not know them: classified again without, and the quad lists of the level with it -- not
                    //  expected to happen: every ACE of a level takes the same route)
                    check(ace_classify(lab, L.u5, L.need, L.list, L.cnt, B, r, r, L.TH, st, want ? L.e16 : nullptr), "ace_classify");
                    const bool again = lvl_done[k][ti];
                    lvl_done[k][ti] = true;
                    lvl_edges[k][ti] = want;
                    if (again) {
                        wq_done[k] = false;
                        wwork_done[k].clear();
                        work_done[k].clear();
                    }
                }
                bool done = false;
                for (int d : work_done[k]) done = done || d == mt;
                if (!done) {
                    check(ace_worklist(L.cnt, ntiles, mt, w.work, w.total, st, w.mode, 32 * L.TH, w.work2, w.total2, w.work3), "ace_worklist");
                    work_done[k].push_back(mt);
                }
                return &w;
            }
        return nullptr;
    }
    // Winograd ACE path: classification of the level (when the interior reduction serves it), boundary-quad lists, task list
    struct WinoPrep { const SeanModel::WinoLevel* L = nullptr; const SeanModel::WinoWork* W = nullptr; const SparseLevel* S = nullptr; bool edges = false; };
    bool wq_done[6] = {};
    bool lvl_edges[6][2] = {};          // the level's interior map carries straight-edge marks (u5 == 253 + e16)
    std::vector<int> wwork_done[6];
    WinoPrep wino_prepare(const AceW& a, const uint8_t* lab, int r) {
        WinoPrep o;
        int k = 0;
        while ((1 << k) < a.res_div) ++k;
        const SeanModel::WinoLevel& L = m.wq_level[k];
        if (!L.qlist) return o;
        const int nrt = (a.C + 15) / 16;
        for (const auto& w : L.works)
            if (w.nrt == nrt) o.W = &w;
        if (!o.W) return o;
        o.L = &L;
        if (m.sparse && r >= m.sparse_min_r) {
            for (int ti = 1; ti >= 0 && !o.S; --ti)
                if (m.sp_level[k][ti].u5) {
                    const SparseLevel& S = m.sp_level[k][ti];
                    if (!lvl_done[k][ti]) {
                        // straight-edge pixels (ace_sparse.h): marked on the levels whose interior pass knows them (128 pixels and more) when
                        // the ACE carries its table -- every ACE of such a level does (res_div <= 4), checked in ace()
                        const bool edges = m.edge && S.e16 && r >= 128 && a.edge_tab && (!a.styled || m.p6) && !m.overlap_on;      // (overlap mode: quad_only interior pass)
                        check(ace_classify(lab, S.u5, S.need, S.list, S.cnt, B, r, r, S.TH, st, edges ? S.e16 : nullptr), "ace_classify");
                        lvl_done[k][ti] = true;
                        lvl_edges[k][ti] = edges;
                    }
                    o.S = &S;
                    o.edges = lvl_edges[k][ti];
                }
        }
        const int ntiles = B * (r / 32) * (r / L.TH);
        if (!wq_done[k]) {
            check(wino_quad_lists(o.S ? o.S->u5 : nullptr, L.qlist, L.qcnt, L.pcnt, B, r, r, L.TH, st), "wino_quad_lists");
            if (L.gq) check(wino_gather_lists(L.qlist, L.qcnt, L.qoff, L.gq, L.gq_n, L.gq_cap, B, r, r, st), "wino_gather_lists");
            if (L.gq && L.chunk_base) check(wino_chunk_base(L.gq_n, B, m.patchbuf ? m.patch_cap_chunks : 0, L.chunk_base, L.patch_mode, st), "wino_chunk_base");
            wq_done[k] = true;
        }
        bool done = false;
        for (int d : wwork_done[k]) done = done || d == nrt;
        if (!done) {
            if (L.gq) check(wino_gather_worklist(L.gq_n, L.pcnt, B, ntiles / B, nrt, o.W->work, o.W->total, st), "wino_gather_worklist");
            else check(wino_ace_worklist(L.qcnt, L.pcnt, ntiles, nrt, o.W->work, o.W->total, st), "wino_ace_worklist");
            wwork_done[k].push_back(nrt);
        }
        return o;
    }
    const uint8_t* labels_at(const uint8_t* full, int res_div) {
        if (res_div == 1) return full;
        int k = 0;
        while ((1 << k) < res_div) ++k;
        return m.lab_r[k];
    }
    // the interior map of an ACE's level once its classification has run in this chunk (else nullptr)
    const uint8_t* level_u5(const AceW& a, int r) const {
        if (!m.sparse || r < m.sparse_min_r) return nullptr;
        int k = 0;
        while ((1 << k) < a.res_div) ++k;
        for (int ti = 1; ti >= 0; --ti)
            if (m.sp_level[k][ti].u5 && lvl_done[k][ti]) return m.sp_level[k][ti].u5;
        return nullptr;
    }
    // classification, quad lists and task lists of every Winograd ACE level up front (they depend on the labels only): the label
    // tables that run ahead on the side stream need the interior maps
    void prepass(const uint8_t* labfull) {
        for (const auto& b : m.blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!a) continue;
                const int r = S / a->res_div;
                if (use_wino_ace(*a, r) && !use_wino4_ace(*a, r)) (void)wino_prepare(*a, labels_at(labfull, a->res_div), r);
            }
    }

    // What an ACE needs that depends on the label map and the style codes only -- never on the activations flowing
    // through the generator: the SPADE hidden activations (label table) and the style LUT (fc_mu + LUT GEMM / GEMV).
    struct AcePrep {
        const float* actv = nullptr;
        const float* lut = nullptr;
        int lut_rs = 1, lut_ns = 0, lut_bs = LABEL_NC;
    };
    // what: bit 0 = style LUT, bit 1 = SPADE hidden activations
    // ONE predicate decides both the layout of the hidden activations (148 planes per sample with the one-hot planes behind the
    // hidden channels) and the kernel that reads them: the level's quad lists and the task list of this row-tile count must have
    // been sized at build time (they exist where (max_size >> k) % 32 == 0; a smaller S can reach a 32-grid level that max_size
    // does not have, e.g. max_size = 96, S = 64 -- ADVICE r04), else the ACE takes the direct kernels and their [B][128][H][W] layout
    bool use_wino_ace(const AceW& a, int r) const {
        if (!(m.wino && !m.use_sh16 && a.spade_wino && r >= 32 && r % 32 == 0)) return false;
        int k = 0;
        while ((1 << k) < a.res_div) ++k;
        const SeanModel::WinoLevel& L = m.wq_level[k];
        if (!L.qlist) return false;
        const int nrt = (a.C + 15) / 16;
        for (const auto& w : L.works)
            if (w.nrt == nrt) return true;
        return false;
    }
    // u5 (Winograd ACE path): the level's interior map when the SPADE-interior reduction serves it -- the label-table kernel then
    // writes the hidden activations only where a boundary quad's patch reads them (nullptr: everywhere)
    // SPADE conv of this ACE as F(4x4,3x3) over every tile of the level (conv_wino4.h): the low-resolution levels, where nearly every
    // tile holds a boundary pixel -- no classification, no interior pass, hidden activations at every pixel
    bool use_wino4_ace(const AceW& a, int r) const {
        return use_wino_ace(a, r) && m.wino >= 2 && a.spade_wino4 && r <= m.wino4_ace_max_r && wino4_ace_supported(r, r, a.C) && (!a.styled || m.wsty4) &&
               (m.wino4_force || m.batch_inv || wino4_pays((long long)B * (r / 32) * (r / 32) * ((a.C + 15) / 16), m.num_cus));      // (few tasks per CU -- single images -- : the gather kernel)
    }
    AcePrep ace_prepare(const AceW& a, const uint8_t* labfull, const float* codes, hipStream_t s, float* actv_buf, float* lut_buf,
                        float* splitk, bool prof, int what = 3, const uint8_t* need = nullptr, const int* tile_cnt = nullptr,
                        const uint8_t* u5 = nullptr, const SeanModel::WinoLevel* PL = nullptr) {      // PL: the level, when its patch source may serve this ACE
        const int r = S / a.res_div;
        const uint8_t* lab = labels_at(labfull, a.res_div);
        AcePrep q;
        q.actv = actv_buf;
        q.lut_ns = 18 * a.C;
        auto tm = [&](double flops, double bytes, auto launch) {
            if (prof) timed(2, flops, bytes, launch);
            else launch();
        };
        if (a.styled && (what & 1)) {
            q.lut = lut_buf;
            // f16x3 path: one extra all-zero column per sample (mu = 0 -> LUT = 0) that taps outside the image point at
            const int bs = m.use_sh16 ? LABEL_NC + 1 : LABEL_NC;
            const int N = B * bs, npad = ((N + 31) / 32) * 32;
            q.lut_bs = bs;
            const double fl = 2.0 * 18 * a.C * STYLE * N, by = 4.0 * (18.0 * a.C * STYLE + (double)STYLE * N + 18.0 * a.C * N);
            if (a.lut_rows && N <= 64 && !m.batch_inv) {
                // interactive batch sizes: P[n][row] = sum_k W[row][k] mu[n][k] as a batched GEMV (weight-bandwidth bound)
                check(fc_mu(codes, a.fcmu_w, a.fcmu_b, m.mu_img, B, npad, s, m.mu_img, 0, bs), "fc_mu");
                tm(fl, by, [&] {
                    check(lut_gemv_mfma(m.mu_img, a.lut_rows, lut_buf, N, 18 * a.C, s), "lut gemv");
                });
            } else if (m.use_sh16) {
                // f16x3 LUT GEMM: 1x1 conv over the [npad/32 x 32] "image" of (sample, label) columns, C4 output
                unsigned* mu_slot = m.amax_slots + 2 * a.index + 1;
                const float* mu = m.mu_img;
                if (m.fcmu_batched) {                    // projected at the start of the chunk, all ACEs in one launch
                    mu = m.mu_all + (size_t)a.index * m.mu_stride;
                } else {
                    for (int pass = 0; pass < 2; ++pass)     // second pass: returns at once unless the first one left the f16 window
                        check(fc_mu(codes, a.fcmu_w, a.fcmu_b, m.mu_img, B, npad, s, nullptr, 1, bs, SH16_ACT_SCALE, mu_slot, pass, m.terms == 2), "fc_mu");
                }
                ConvParams p{};
                p.in = mu;
                p.wpk = a.lut_wpk;
                p.out = lut_buf;
                p.B = 1;
                p.Cin = STYLE;
                p.H = npad / 32;
                p.W = 32;
                p.Mrows = 18 * a.C;
                p.pad = -1;
                p.act = ACT_NONE;
                p.terms = m.terms;
                p.wscale = a.lut_wscale;
                p.in_scale_inv = 1.f / SH16_ACT_SCALE;
                p.in_amax = mu_slot;
                p.out_mul = a.out_scale;           // the ACE epilogue takes the LUT pre-multiplied by its output scale
                p.partial = splitk;                // K = 512 in 32 chunks on few tiles (C <= 512): split-K fills the chip
                p.partial_cap = m.splitk_cap;
                tm(fl, by, [&] { check(conv_sh16_plain(p, 1, s), "lut gemm"); });
                q.lut_rs = npad;
                q.lut_ns = 4;
            } else {
                const float* mu = m.mu_img;
                if (m.fcmu_batched && (N > 64 || m.batch_inv)) mu = m.mu_all + (size_t)a.index * m.mu_stride;      // projected at the start of the chunk (generate())
                else check(fc_mu(codes, a.fcmu_w, a.fcmu_b, m.mu_img, B, npad, s), "fc_mu");
                ConvParams p{};
                p.in = mu;
                p.wpk = a.lut_wpk;
                p.out = lut_buf;
                p.B = 1;
                p.Cin = STYLE;
                p.H = npad / 32;
                p.W = 32;
                p.Mrows = 18 * a.C;
                p.npix_valid = N;
                p.pad = -1;
                tm(fl, by, [&] { check(conv_nhwc1x1(p, s), "lut gemm"); });
            }
        }
        if (!(what & 2)) return q;
        // a buffer that held the padded planes of a Winograd level (zero columns cleared once per geometry) and now receives the
        // direct [B][HID][r][r] layout -- the same ACE at another image size on a run-ahead handle -- must have its pads cleared again
        // before the next padded use: the direct kernels write over them
        if (m.use_sh16 || !use_wino_ace(a, r)) m.pad_state.erase(actv_buf);
        if (m.use_sh16)
            check(onehot_conv3x3_sh16(lab, a.actv_table, a.actv_bias, actv_buf, B, r, r, HID, 1, a.actv_scale, s, m.terms == 2, need, tile_cnt,
                                      (m.dbg & 33554432) ? 1 : 0), "mlp_shared");
        else if (use_wino_ace(a, r)) {
            // Winograd ACE path: hidden activations (+ the one-hot planes of a styled ACE, which feed the style k-steps) in one
            // persistent pass into the padded planes the ACE kernels read (conv_wino.h WINO_AXOFF), whole 64-byte groups of pixels
            // that a boundary quad's patch touches only (sean.hidden_wq = 0: every pixel)
            // the zero columns left and right of the image: the kernel never writes them, so they are cleared whenever the buffer
            // changes its geometry (first use, another image size) -- per-pixel pad stores cost 200 us per 512^2 launch
            const int kout = a.styled ? HID + 20 : HID;
            const long long geo = ((long long)r << 20) | ((long long)kout << 8) | 1;
            auto it = m.pad_state.find(actv_buf);
            if (it == m.pad_state.end() || it->second != geo) {
                check(hipMemsetAsync(actv_buf, 0, (size_t)m.max_batch * kout * r * wino_apitch(r) * sizeof(float), s), "hidden activations: zero pads");
                m.pad_state[actv_buf] = geo;
            }
            const bool pm = PL && PL->gq && PL->patch_mode && m.patchbuf;
            check(spade_hidden_wq(lab, m.hidden_wq ? u5 : nullptr, a.actv_table, a.actv_bias, actv_buf, B, r, r, kout, a.styled ? 1 : 0, s,
                                  wino_apitch(r), WINO_AXOFF, pm ? PL->patch_mode : nullptr), "mlp_shared (boundary-quad patches)");
            // few, scattered boundary quads: their patches pre-gathered instead (one of the two kernels returns at once: device flag)
            if (pm) check(spade_hidden_patch(lab, PL->gq, PL->gq_n, PL->gq_cap, PL->chunk_base, PL->patch_mode, a.actv_table, a.actv_bias, m.patchbuf,
                                             B, r, r, kout, s), "mlp_shared (pre-gathered patches)");
        } else
            check(onehot_conv3x3(lab, a.actv_table, a.actv_bias, actv_buf, B, r, r, HID, 1, s, 0, need), "mlp_shared");
        return q;
    }

    // Small jobs (interactive renders: the chip is mostly empty) run every ACE's prepare step AHEAD on the model's side
    // stream, into per-ACE buffers, while the main stream walks the dependent chain of convs; an event per ACE joins them.
    // full = false (large jobs): only the style LUT builds run ahead; the label-table kernels stay inline.
    bool ahead = false, ahead_luts = false;
    std::vector<AcePrep> prepared;
    bool luts_ready = false;
    void lut_entry(const AceW& a, AcePrep& e) const {
        e.lut = m.lut_ahead[a.index];
        e.lut_rs = 1;
        e.lut_ns = 18 * a.C;
        e.lut_bs = LABEL_NC;
    }
    void prepare_all_ahead(const uint8_t* labfull, const float* codes, bool full) {
        ahead = full;
        ahead_luts = !full;
        prepared.assign(m.n_aces, AcePrep());
        check(hipEventRecord(m.ev_fork, st), "fork");
        check(hipStreamWaitEvent(m.side, m.ev_fork, 0), "fork wait");
        for (const auto& b : m.blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (!a || (!full && !a->styled)) continue;
                const int ra = S / a->res_div;
                AcePrep e = ace_prepare(*a, labfull, codes, m.side, m.actv_ahead[a->index], m.lut_ahead[a->index], m.splitk_side, false,
                                        full ? (luts_ready ? 2 : 3) : 1, nullptr, nullptr,
                                        (use_wino_ace(*a, ra) && !use_wino4_ace(*a, ra)) ? level_u5(*a, ra) : nullptr);
                if (luts_ready && a->styled) lut_entry(*a, e);          // (the grouped launch on the main stream built it)
                prepared[a->index] = e;
                check(hipEventRecord(m.ev_join[a->index], m.side), "join record");
            }
    }

    // Exact-f32 path, more than 64 (sample, label) columns: the style LUTs P[(sample, label)][18 C] of ALL styled ACEs from one
    // grouped GEMM launch on the main stream (conv_pw.h; 15 separate launches of the generic 1x1 kernel took 3.6 ms of GPU time per
    // step on the side stream and delayed the persistent conv kernels they shared CUs with by 1.8 ms; inline 2.8 ms).  Operands
    // swapped: GEMM rows = the (sample, label) columns (A = the projected codes, packed by fc_mu_batched), "pixels" = the 18 C rows
    // of conv_gamma / conv_beta (normalization.py:172-173), so the output is the [n][18 C] layout the consumers read.
    void luts_grouped() {
        const int N = B * LABEL_NC;
        PwParams q{};
        q.Cin = STYLE;
        q.Cout = N;
        q.groups = static_cast<const PwGroup*>(m.lut_groups);
        q.ngroups = m.lut_ngroups;
        const int nrg = ((N + 31) / 32 + 3) / 4;
        timed(2, 2.0 * m.lut_group_rows * STYLE * N, 4.0 * (m.lut_group_rows * STYLE + m.lut_group_rows * N + (double)m.lut_ngroups * STYLE * N),
              [&] { check(conv_pw_grouped(q, m.lut_group_tiles * nrg, st), "style LUTs (grouped GEMM)"); });
        prepared.assign(m.n_aces, AcePrep());
        for (const auto& b : m.blocks)
            for (const AceW* a : {b.learned ? &b.ace_s : nullptr, &b.ace_0, &b.ace_1}) {
                if (a && a->styled) lut_entry(*a, prepared[a->index]);
            }
        luts_ready = true;
    }

    // one ACE: SPADE hidden activations (label LUT) -> fused gamma/beta conv + modulation -> h
    void ace(const AceW& a, const uint8_t* labfull, const float* codes, const float* noise, size_t nf, size_t noff,
             const float* x, int x_up, int act, float* hout) {
        const int r = S / a.res_div;
        const uint8_t* lab = labels_at(labfull, a.res_div);
        const double npix = (double)B * r * r;
        // exact SPADE-interior reduction (ace_sparse.h): classification of the level + work list of this layer's row tiles.
        // f16x3 path: tile-skip mode, honoured by the wave-specialised kernel only (conv_sh16.h)
        const bool wino_ace = use_wino_ace(a, r), f4_ace = use_wino4_ace(a, r);
        WinoPrep wp;
        if (wino_ace && !f4_ace) wp = wino_prepare(a, lab, r);
        const SparseWork* sw = (wino_ace && (wp.L || f4_ace)) ? nullptr : sparse_prepare(a, lab, r);
        if (sw && m.use_sh16) {
            ConvParams t{};
            t.C = a.C;
            t.W = t.H = r;
            t.B = B;
            t.Cin = HID;
            t.dbg = m.dbg;
            if (!sh16_ace_uses_ws(t)) sw = nullptr;
        }
        const SparseLevel* SL = nullptr;
        if (sw) {
            int lk = 0;
            while ((1 << lk) < a.res_div) ++lk;
            SL = &m.sp_level[lk][sw->TH == 16 ? 1 : 0];
        }
        // the label-table kernel only writes the hidden activations the conv will read
        // (exact-f32 path: skipping pixels of the NCHW planes made the kernel slower at every granularity tried -- it is bound by
        // its LDS table reads, not by the writes; the map stays available behind sean.dbg bit 131072)
        const uint8_t* need0 = (SL && !m.use_sh16 && (m.dbg & 131072)) ? SL->need : nullptr;
        const uint8_t* need = need0;
        const bool compact = SL && m.use_sh16 && sw->mode >= 2;      // f16x3: pixel-level compaction inside the ws kernel
        if (compact) need = SL->need;
        const int* tile_cnt = (SL && m.use_sh16 && !compact) ? SL->cnt : nullptr;
        AcePrep q;
        const uint8_t* u5 = (wino_ace && wp.L && wp.S) ? wp.S->u5 : nullptr;      // (F(4x4,3x3) levels: every pixel is read)
        // the level whose pre-gathered patches may serve this ACE (gather mode, not run-ahead: the ahead buffers are per ACE)
        const SeanModel::WinoLevel* patch_level = (wino_ace && !f4_ace && wp.L && wp.L->gq && !ahead && m.patchbuf && !overlap) ? wp.L : nullptr;
        float* abuf = m.actv;                      // hidden activations: the Winograd ACE levels own padded buffers (pads zeroed once per size)
        if (wino_ace) {
            int lk = 0;
            while ((1 << lk) < a.res_div) ++lk;
            if (m.actv_lvl[lk]) abuf = m.actv_lvl[lk];
        }
        if (ahead) {
            q = prepared[a.index];
            check(hipStreamWaitEvent(st, m.ev_join[a.index], 0), "join wait");
        } else if (luts_ready && a.styled) {
            (void)ace_prepare(a, labfull, codes, st, abuf, nullptr, m.splitk_ws, true, 2, need, tile_cnt, u5, patch_level);     // label table inline
            q = prepared[a.index];
            q.actv = abuf;
        } else if (ahead_luts && a.styled) {
            (void)ace_prepare(a, labfull, codes, st, abuf, nullptr, m.splitk_ws, true, 2, need, tile_cnt, u5, patch_level);     // label table inline
            q = prepared[a.index];
            q.actv = abuf;
            check(hipStreamWaitEvent(st, m.ev_join[a.index], 0), "join wait");
        } else {
            q = ace_prepare(a, labfull, codes, st, abuf, m.lut, m.splitk_ws, true, 3, need, tile_cnt, u5, patch_level);
        }
        if (f4_ace) {
            Wino4AceParams w{};
            w.actv = q.actv;
            w.wpk = a.spade_wino4;
            w.wsty = (a.styled && q.lut) ? m.wsty4 : nullptr;
            w.out = hout;
            w.x = x;
            w.x_up = x_up;
            w.act = act;
            w.B = B;
            w.C = a.C;
            w.H = r;
            w.W = r;
            w.bias_g = a.bias_g;
            w.bias_b = a.bias_b;
            w.bn_a = a.bn_a;
            w.bn_d = a.bn_d;
            w.nv = a.nv;
            w.noise = noise + noff;
            w.noise_bstride = (long long)nf;
            const int ktot = w.wsty ? 152 : 128;             // (38 / 32 k-steps of four channels)
            next_flops_exec = 2.0 * 32.0 * ((a.C + 15) / 16) * ktot * 36.0 * npix / 16.0;
            // 2 C GEMM rows over 128 (+ 20) input planes: the input transform once for all C / 8 row tiles (conv_wino4v.h)
            const bool vroute = m.wino4v && wino4v_pays(2 * a.C, r) && m.wino4v_fits(B, r, w.wsty ? 38 : 32);
            timed(1, 2.0 * 2 * a.C * HID * 9 * npix, 4.0 * (npix * (HID + (a.styled ? 20 : 0)) + npix * a.C / (x_up ? 4.0 : 1.0) + npix * a.C), [&] {
                if (w.wsty) check(wino4_style_pack(q.lut, m.wsty4, B, a.C, st), "wino4_style_pack");
                if (vroute) {
                    Wino4vPackParams vp{};
                    vp.in = q.actv;
                    vp.v = m.vbuf;
                    vp.B = B;
                    vp.K = HID + (a.styled ? 20 : 0);       // (the buffer's plane count: ace_prepare; an ACE without LUT reads the first 128 only)
                    vp.H = vp.W = r;
                    vp.nks = w.wsty ? 38 : 32;
                    vp.pitch = wino_apitch(r);
                    vp.xoff = WINO_AXOFF;
                    vp.padded = 1;
                    check(wino4v_pack(vp, st), "hidden activation transform (winograd F(4x4,3x3))");
                    w.v = m.vbuf;
                    check(conv_wino4v_ace(w, st), "spade conv (winograd F(4x4,3x3), every tile, pre-transformed input)");
                } else
                    check(conv_wino4_ace(w, st), "spade conv (winograd F(4x4,3x3), every tile)");
            });
            return;
        }
        if (wino_ace && wp.L) {
            const double xpp = 4.0 * a.C / (x_up ? 4.0 : 1.0), opp = 4.0 * a.C;
            const int ktot = HID + (a.styled ? 20 : 0);
            if (wp.S) {          // interior pixels: elementwise with the per-(sample, label) gamma/beta rows (ace_sparse.h)
                AceInteriorParams ip{};
                ip.x = x;
                ip.out = hout;
                ip.u5 = wp.S->u5;
                ip.cnt = wp.S->cnt;
                ip.gtab = m.gtab;
                ip.bn_a = a.bn_a;
                ip.bn_d = a.bn_d;
                ip.nv = a.nv;
                ip.noise = noise + noff;
                ip.noise_bstride = (long long)nf;
                ip.B = B;
                ip.C = a.C;
                ip.H = r;
                ip.W = r;
                ip.x_up = x_up;
                ip.act = act;
                ip.variant = 0;
                // four pixels per thread (16-byte stores) from 128 pixels of width: 530 -> 420 us on the up-sampled 512^2 launches
                // (tools/interior_bench.hip); level in the 100 ms step of round 3, measurable in this one
                ip.impl = r >= 128 ? 2 : 0;
                if (wp.edges) {         // straight-edge pixels: table row of the code + three column / row sums of the style LUT (built below)
                    if (!a.edge_tab || r < 128) check(hipErrorInvalidValue, "straight-edge marks on a level whose ACE has no table");
                    ip.e16 = wp.S->e16;
                    ip.etab = a.edge_tab;
                    ip.p6 = (a.styled && q.lut) ? m.p6 : nullptr;
                }
                // (tile4 kernel, r >= 128: 0 = whole 128-byte lines that hold an interior pixel, ace_sparse.hip; sean.dbg bit 134217728: the
                //  block rule of rounds 3-5 for A/B)
                ip.fill_min = (x_up && r < 128) ? 257 : ((r >= 128 && !(m.dbg & 134217728)) ? 0 : 128);
                if (overlap) {
                    // beside the boundary conv of the same ACE (disjoint output pixels -- so no block may fill its boundary pixels),
                    // on the few CUs of the side stream; the consumer of `hout` waits for ev_int below
                    ip.fill_min = 257;
                    ip.quad_only = 1;
                    ip.gtab = m.gtab_side;
                    check(hipEventRecord(m.ev_x[a.index], st), "x ready");
                    check(hipStreamWaitEvent(m.side_int, m.ev_x[a.index], 0), "x ready wait");
                    check(ace_gtable(a.bias_g, a.bias_b, a.gconst, q.lut, q.lut_rs, q.lut_ns, q.lut_bs, 1.f, m.gtab_side, B, a.C, m.side_int), "ace_gtable");
                    check(ace_interior_f32(ip, m.side_int), "ace interior");
                    check(hipEventRecord(m.ev_int[a.index], m.side_int), "interior done");
                } else {
                    timed(3, 0.0, 0.0, wp.W->total + 4, 0.0, xpp + opp + 5.0, 4.0 * 19 * 2 * a.C * B, npix, [&] {
                        check(ace_gtable(a.bias_g, a.bias_b, a.gconst, q.lut, q.lut_rs, q.lut_ns, q.lut_bs, 1.f, m.gtab, B, a.C, st), "ace_gtable");
                        if (ip.p6) check(ace_p6table(q.lut, q.lut_rs, q.lut_ns, q.lut_bs, 1.f, m.p6, B, a.C, st), "ace_p6table");
                        check(ace_interior_f32(ip, st), "ace interior");
                    });
                }
            }
            WinoAceParams w{};
            w.actv = q.actv;
            w.wpk = a.spade_wino;
            w.wsty = (a.styled && q.lut) ? m.wsty : nullptr;
            w.out = hout;
            w.x = x;
            w.x_up = x_up;
            w.act = act;
            w.B = B;
            w.C = a.C;
            w.H = r;
            w.W = r;
            w.bias_g = a.bias_g;
            w.bias_b = a.bias_b;
            w.bn_a = a.bn_a;
            w.bn_d = a.bn_d;
            w.nv = a.nv;
            w.noise = noise + noff;
            w.noise_bstride = (long long)nf;
            w.patch = patch_level ? m.patchbuf : nullptr;
            w.chunk_base = wp.L->chunk_base;
            w.patch_mode = wp.L->patch_mode;
            w.qlist = wp.L->qlist;
            w.TH = wp.L->TH;
            w.qcnt = wp.L->qcnt;
            w.work = wp.W->work;
            w.total = wp.W->total;
            w.zero = m.zero_page;
            w.gq = wp.L->gq;
            w.gq_n = wp.L->gq_n;
            w.gq_cap = wp.L->gq_cap;
            w.claim = w.gq ? next_claim() : nullptr;
            // executed FLOPs = wave tasks x (32 rows x 16 quads x 16 positions x K) x 2; dense = the direct conv over every pixel
            timed(1, 2.0 * 2 * a.C * HID * 9 * npix, 0.0, wp.W->total + 4, 2.0 * 32 * 16 * 16 * ktot, 4.0 * ktot + xpp + opp,
                  4.0 * 2.0 * a.C * ktot * 16, npix, [&] {
                      if (w.wsty) check(wino_style_pack(q.lut, m.wsty, B, a.C, st), "wino_style_pack");
                      check(conv_wino_ace(w, st), "spade conv (winograd, boundary quads)");
                  });
            if (overlap && wp.S) check(hipStreamWaitEvent(st, m.ev_int[a.index], 0), "interior done wait");
            return;
        }
        ConvParams p{};
        p.in = q.actv;
        p.wpk = a.spade_wpk;
        p.out = hout;
        p.B = B;
        p.Cin = HID;
        p.H = r;
        p.W = r;
        p.Mrows = 2 * a.C;
        p.x = x;
        p.x_up = x_up;
        p.C = a.C;
        p.bias_g = a.bias_g;
        p.bias_b = a.bias_b;
        p.bn_a = a.bn_a;
        p.bn_d = a.bn_d;
        p.nv = a.nv;
        p.noise = noise + noff;
        p.noise_bstride = (long long)nf;
        p.lab = lab;
        p.lut = q.lut;
        p.lut_rs = q.lut_rs;
        p.lut_ns = q.lut_ns;
        p.lut_bs = q.lut_bs;
        p.act = act;
        p.pad = -1;
        p.dbg = m.dbg;
        p.terms = m.terms;
        p.wscale = a.spade_wscale;
        p.in_scale_inv = 1.f / a.actv_scale;
        p.out_scale = a.out_scale;
        const double xin = npix * a.C / (x_up ? 4.0 : 1.0);
        p.out_amax = m.amax_slots + 2 * a.index;
        if ((m.dbg & 256) && a.index == m.dbg_sel) p.partial = m.splitk_ws;      // cycle stamps of this launch (profiling)
        else p.dbg &= ~256;
        if (sw) {
            // interior pixels: elementwise with the per-(sample, label) gamma/beta rows; the others: conv over the compacted
            // boundary pixels (exact-f32 path) / over the tiles that hold a boundary pixel (f16x3 path)
            const SparseLevel& L = *SL;
            const double xpp = 4.0 * a.C / (x_up ? 4.0 : 1.0), opp = 4.0 * a.C;
            AceInteriorParams ip{};
            ip.x = x;
            ip.out = hout;
            ip.u5 = L.u5;
            ip.cnt = L.cnt;
            ip.gtab = m.gtab;
            ip.bn_a = a.bn_a;
            ip.bn_d = a.bn_d;
            ip.nv = a.nv;
            ip.noise = noise + noff;
            ip.noise_bstride = (long long)nf;
            ip.B = B;
            ip.C = a.C;
            ip.H = r;
            ip.W = r;
            ip.x_up = x_up;
            ip.act = act;
            ip.out_scale = a.out_scale;
            ip.out_amax = m.amax_slots + 2 * a.index;
            ip.bf16 = m.terms == 2;
            ip.single = (m.use_sh16 && m.terms != 3 && !(m.dbg & 268435456)) ? 1 : 0;      // (dbg bit: A/B)
            ip.variant = m.use_sh16 ? (compact ? 1 : 0) : (CH_ABL(m.dbg & 65536) ? 1 : (CH_ABL(m.dbg & 1048576) ? 2 : 0));
            // blocks of 32 x 8 pixels; mostly-interior blocks write every pixel (the conv below overwrites the boundary pixels).
            // Exact-f32 pass: filling pays only where x is read at full size (measured, tools/interior_bench.hip).
            // dbg bit 2097152: the row-shaped kernels of the first version (A/B)
            ip.impl = CH_ABL(m.dbg & 2097152) ? 1 : 0;
            ip.fill_min = m.use_sh16 ? 128 : (x_up ? 257 : 128);
            if (m.use_sh16 && compact && lvl_edges[lk_of(a)][sw->TH == 16 ? 1 : 0]) {      // straight-edge pixels served by the interior pass
                ip.e16 = L.e16;
                ip.etab = a.edge_tab;
                ip.p6 = (a.styled && q.lut) ? m.p6 : nullptr;
            }
            // (impl 2, four pixels per thread: 10 % ahead at 512^2 in tools/interior_bench.hip, no difference in the generator
            // step -- 159.7 vs 159.9 images/s -- so the one-pixel kernel stays; dbg bit 16777216 selects it)
            if (!m.use_sh16 && CH_ABL(m.dbg & 16777216) && r >= 128) {
                ip.impl = 2;
                ip.fill_min = 128;
            }
            p.sp_list = (m.use_sh16 && !compact) ? nullptr : L.list;      // f16x3: the lists request the compacting kernel
            p.sp_cnt = L.cnt;
            p.sp_work = sw->work;
            p.sp_total = sw->total;
            p.sp_work2 = sw->work2;
            p.sp_total2 = sw->total2;
            p.sp_work3 = sw->work3;
            timed(3, 0.0, 0.0, sw->total, 0.0, xpp + opp + 5.0, 4.0 * 19 * 2 * a.C * B, npix, [&] {
                // (the f16x3 LUT is stored pre-multiplied by the ACE output scale)
                check(ace_gtable(a.bias_g, a.bias_b, a.gconst, q.lut, q.lut_rs, q.lut_ns, q.lut_bs, m.use_sh16 ? 1.f / a.out_scale : 1.f,
                                 m.gtab, B, a.C, st), "ace_gtable");
                if (ip.p6) check(ace_p6table(q.lut, q.lut_rs, q.lut_ns, q.lut_bs, 1.f / a.out_scale, m.p6, B, a.C, st), "ace_p6table");
                check(m.use_sh16 ? ace_interior_sh16(ip, st) : ace_interior_f32(ip, st), "ace interior");
            });
            timed(1, 2.0 * 2 * a.C * HID * 9 * npix, 0.0, sw->total, 32.0 * 64 * 2.0 * HID * 9, 4.0 * HID + xpp + opp,
                  4.0 * 2.0 * a.C * HID * 9, npix, [&] {
                      if (m.use_sh16) check(conv_sh16_ace(p, st), "spade conv (tiles with boundary pixels)");
                      else check(conv_ace_sparse(p, L.TH, st), "spade conv (boundary pixels)");
                  });
            if (m.use_sh16) {      // second passes: return at once unless the recorded maximum left the f16 window (sh16.h)
                ip.pass = 1;
                check(ace_interior_sh16(ip, st), "ace interior (second pass)");
                p.pass = 1;
                check(conv_sh16_ace(p, st), "spade conv (second pass)");
            }
            return;
        }
        if (!m.use_sh16 && m.gb_small && !m.batch_inv) {
            // Tiny levels at small batches (one image at 16 x 16: 16 blocks of the fused kernel, 8 % of the CUs, 200 us): the SPADE conv as a
            // PLAIN conv over the same packed image -- which splits K over blocks when its grid is small (conv_mfma.h launch_conv) -- into
            // a scratch of gamma | beta sums, then ace_finish_f32 (style-LUT gathers, biases, modulation).  Same sums per element, the
            // split-K slabs added in order: deterministic.
            const int rowsP = ((a.C + 31) / 32) * 64;
            const long long tiles = ((long long)B * r * r + 255) / 256;
            if (((rowsP + 127) / 128) * tiles < 128 && (long long)rowsP * B * r * r <= m.gb_small_cap) {
                ConvParams c = p;
                c.out = m.gb_small;
                c.Mrows = rowsP;
                c.bias = nullptr;
                c.res = nullptr;
                c.act = ACT_NONE;
                c.partial = m.splitk_ws;
                c.partial_cap = m.splitk_cap;
                c.dbg &= ~256;
                timed(1, 2.0 * 2 * a.C * HID * 9 * npix, 4.0 * (npix * HID + xin + npix * a.C + 2.0 * a.C * HID * 9), [&] {
                    check(conv_plain3(c, st), "spade conv (plain, split-K)");
                    check(ace_finish_f32(m.gb_small, rowsP, x, x_up, a.bias_g, a.bias_b, a.bn_a, a.bn_d, a.nv, noise + noff, (long long)nf, lab, q.lut,
                                         static_cast<float*>(hout), B, a.C, r, r, act, st), "ace finish");
                });
                return;
            }
        }
        timed(1, 2.0 * 2 * a.C * HID * 9 * npix, 4.0 * (npix * HID + xin + npix * a.C + 2.0 * a.C * HID * 9), [&] {
            if (!m.use_sh16) {
                check(conv_ace(p, st), "spade conv");
                return;
            }
            check(conv_sh16_ace(p, st), "spade conv");
            p.pass = 1;        // returns at once unless the recorded maximum left the f16 window (sh16.h)
            check(conv_sh16_ace(p, st), "spade conv (second pass)");
        });
    }

    // fuse: optional (weights, input) of a 1x1 conv added into this 3x3 conv's accumulators (f16x3 path, W >= 32, no split-K)
    bool can_fuse_1x1(const ConvW& w, int r) const {
        return !(m.dbg & 32) && w.KS == 3 && r >= 32 &&
               !(((w.Cout + 63) / 64) * (((long long)B * r * r + 511) / 512) < 192);
    }
    bool use_wino(const ConvW& w, int r) const {
        return m.wino && !m.use_sh16 && w.wino && w.KS == 3 && (wino_supported(r, r, w.Cin) || (!m.batch_inv && wino_supported_pair16(B, r, r, w.Cin)));
    }
    // `prod` / `prod2`: the ACEs that wrote `in` / `in2` (their slots hold the scale in effect)
    void conv(const ConvW& w, const float* in, const AceW& prod, float* out, int r, const float* res, int res_up,
              const ConvW* w2 = nullptr, const float* in2 = nullptr, const AceW* prod2 = nullptr) {
        ConvParams p{};
        if (w2) {
            p.in2 = in2;
            p.wpk2 = w2->wpk;
            p.Cin2 = w2->Cin;
            p.in2_amax = m.amax_slots + 2 * prod2->index;
        }
        p.in_amax = m.amax_slots + 2 * prod.index;
        p.in = in;
        p.wpk = w.wpk;
        p.out = out;
        p.B = B;
        p.Cin = w.Cin;
        p.H = r;
        p.W = r;
        p.Mrows = w.Cout;
        p.bias = w.bias;
        p.res = res;
        p.res_up = res_up;
        p.act = ACT_NONE;
        p.pad = -1;
        p.dbg = m.dbg;
        p.terms = m.terms;
        p.wscale = w.wscale;                       // shared with w2's rows when a 1x1 operand is fused (build())
        p.in_scale_inv = 1.f / prod.out_scale;     // inputs are ACE outputs (a fused second operand: see sh16_in_scale_inv)
        p.partial = m.batch_inv ? nullptr : m.splitk_ws;      // (split-K follows the grid size, i.e. the batch)
        p.partial_cap = m.splitk_cap;
        p.mtiles_hint_small = (((w.Cout + 63) / 64) * (((long long)B * r * r + 511) / 512) < 192) ? 1 : 0;
        const double npix = (double)B * r * r, k2 = w.KS * w.KS, cin2 = w2 ? w2->Cin : 0;
        if (m.wino >= 2 && !m.use_sh16 && w.wino4 && w.KS == 3 && !w2 && wino4_supported(r, r, w.Cin) &&
            (!use_wino(w, r) || m.wino4_force || m.batch_inv || wino4_pays((long long)B * (r / 32) * (r / 32) * ((w.Cout + 31) / 32), m.num_cus))) {
            // Winograd F(4x4,3x3) on the exact-f32 matrix cores: 36 MFMA products per 4 x 4 tile and channel instead of 144
            Wino4Params q{};
            q.in = in;
            q.wpk = w.wino4;
            q.out = out;
            q.B = B;
            q.Cin = w.Cin;
            q.Cout = w.Cout;
            q.H = r;
            q.W = r;
            q.bias = w.bias;
            q.res = res;
            q.res_up = res_up;
            next_flops_exec = 2.0 * w.Cout * w.Cin * 36.0 * npix / 16.0;
            // many GEMM rows on a small level: the input transform once, in its own pass, instead of in every row tile (conv_wino4v.h)
            const bool vroute = m.wino4v && wino4v_pays(w.Cout, r) && m.wino4v_fits(B, r, w.Cin / 4);
            timed(0, 2.0 * w.Cout * w.Cin * 9.0 * npix,
                  4.0 * (npix * w.Cin + npix * w.Cout * (res ? 2.0 : 1.0) + (double)w.Cout * w.Cin * 36.0), [&] {
                      if (vroute) {
                          Wino4vPackParams vp{};
                          vp.in = in;
                          vp.v = m.vbuf;
                          vp.B = B;
                          vp.K = w.Cin;
                          vp.H = vp.W = vp.pitch = r;
                          vp.nks = w.Cin / 4;
                          check(wino4v_pack(vp, st), "conv input transform (winograd F(4x4,3x3))");
                          q.v = m.vbuf;
                          check(conv_wino4v_plain(q, st), "conv (winograd F(4x4,3x3), pre-transformed input)");
                      } else
                          check(conv_wino4_plain(q, st), "conv (winograd F(4x4,3x3))");
                  });
            return;
        }
        if (use_wino(w, r) && !w2 && !(r == 16 && res_up)) {      // (16 x 16 sample pairs: residual at the same size only)
            // Winograd F(2x2,3x3) on the exact-f32 matrix cores: 16 MFMA products per quad and channel instead of 36
            WinoParams q{};
            q.in = in;
            q.wpk = w.wino;
            q.out = out;
            q.B = B;
            q.Cin = w.Cin;
            q.Cout = w.Cout;
            q.H = r;
            q.W = r;
            q.bias = w.bias;
            q.res = res;
            q.res_up = res_up;
            q.act = ACT_NONE;
            q.zero = m.zero_page;
            q.claim = next_claim();
            q.partial = m.batch_inv ? nullptr : m.splitk_ws;               // (launches with far fewer tasks than CUs split K: conv_wino_plain)
            q.partial_cap = m.splitk_cap;
            next_flops_exec = 2.0 * w.Cout * w.Cin * 16.0 * npix / 4.0;
            timed(0, 2.0 * w.Cout * w.Cin * 9.0 * npix,
                  4.0 * (npix * w.Cin + npix * w.Cout * (res ? 2.0 : 1.0) + (double)w.Cout * w.Cin * 16.0),
                  [&] { check(conv_wino_plain(q, st), "conv (winograd)"); });
            return;
        }
        if (m.wino && !m.use_sh16 && w.pw && w.KS == 1 && !w2 && !res && !w.bias && pw_supported(w.Cin, w.Cout, r * r)) {
            PwParams q{};         // the learned shortcut conv_s: dedicated 1x1 kernel (conv_pw.h)
            q.in = in;
            q.wpk = w.pw;
            q.out = out;
            q.B = B;
            q.Cin = w.Cin;
            q.Cout = w.Cout;
            q.HW = r * r;
            timed(0, 2.0 * w.Cout * w.Cin * npix, 4.0 * (npix * w.Cin + npix * w.Cout + (double)w.Cout * w.Cin),
                  [&] { check(conv_pw(q, st), "conv 1x1"); });
            return;
        }
        timed(0, 2.0 * w.Cout * (w.Cin * k2 + cin2) * npix,
              4.0 * (npix * (w.Cin + cin2) + npix * w.Cout * (res ? 2.0 : 1.0) + (double)w.Cout * (w.Cin * k2 + cin2)), [&] {
                  if (m.use_sh16) check(conv_sh16_plain(p, w.KS, st), "conv");
                  else check(w.KS == 3 ? conv_plain3(p, st) : conv_plain1(p, st), "conv");
              });
    }
};

}  // namespace

// seed of the planes of the batch chunk that starts at sample `bo` (ch_sean_generate with noise == NULL, ch_sean_draw_noise)
static uint64_t chunk_seed(uint64_t seed, int bo) { return seed + 0x632BE59BD9B4E019ull * (uint64_t)(bo + 1); }

std::string SeanModel::draw_noise(uint64_t seed, float* out, int Btot, int S, hipStream_t st) {
    if (blocks.empty()) return "model not finalized";
    if (S % 32 != 0 || S < 32 || S > max_size) return "S must be a multiple of 32 and <= max_size";
    const size_t nf = noise_floats(S);
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        if (gen_noise(out + (size_t)bo * nf, (long long)B * nf, chunk_seed(seed, bo), st) != hipSuccess) return "gen_noise failed";
    }
    return "";
}

std::string SeanModel::generate(const uint8_t* labels, const float* codes, const float* noise, uint64_t seed,
                                float* out, int Btot, int S, hipStream_t st) {
    if (blocks.empty()) return "model not finalized";
    if (S % 32 != 0 || S < 32 || S > max_size) return "S must be a multiple of 32 and <= max_size";
    if (Btot < 1) return "B must be >= 1";
    const size_t nf = noise_floats(S);
    // overlap handles: everything runs on the internal main stream (the CU-masked side streams are blocking streams: work on the
    // caller's stream, if that is the NULL stream, would serialise with them), forked from and joined to the caller's stream
    hipStream_t st_user = st;
    if (overlap_on) {
        if (hipEventRecord(ev_in, st_user) != hipSuccess || hipStreamWaitEvent(main_i, ev_in, 0) != hipSuccess) return "overlap mode: fork failed";
        st = main_i;
    }
    auto join = [&](const std::string& e) {
        if (overlap_on && (hipEventRecord(ev_out, main_i) != hipSuccess || hipStreamWaitEvent(st_user, ev_out, 0) != hipSuccess) && e.empty())
            return std::string("overlap mode: join failed");
        return e;
    };
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        Runner R(*this, st, B, S);
        if (claim_pool) R.check(hipMemsetAsync(claim_pool, 0, (size_t)CLAIM_SLOTS * CLAIM_WORDS * sizeof(unsigned), st), "claim counters");
        const uint8_t* lab = labels + (size_t)bo * S * S;
        const float* cd = codes + (size_t)bo * LABEL_NC * STYLE;
        const float* nz;
        if (noise) {
            nz = noise + (size_t)bo * nf;
        } else {
            R.check(gen_noise(noise_ws, (long long)B * nf, chunk_seed(seed, bo), st), "gen_noise");
            nz = noise_ws;
        }
        for (int k = 1; k <= 5; ++k) R.check(label_downsample(lab, lab_r[k], B, S, S >> k, st), "label_downsample");
        if (use_sh16) R.check(hipMemsetAsync(amax_slots, 0, 64 * sizeof(unsigned), st), "amax slots");
        if (fcmu_batched && !use_sh16) {
            if (B * LABEL_NC > 64 || batch_inv) {     // (smaller batches take the GEMV branch of ace_prepare, which projects per ACE)
                // grouped LUT build: the projections are its A operand, written in fragment order (sh16 = 2: pack_pw_A layout)
                const bool grouped = lut_groups && lut_ngroups > 0;
                R.check(fc_mu_batched(cd, fcmu_w_ptrs, fcmu_b_ptrs, mu_all, mu_stride, n_aces, B, ((B * LABEL_NC + 31) / 32) * 32, LABEL_NC, 1.f,
                                      nullptr, 0, 0, st, grouped ? 2 : 0), "fc_mu (all ACEs)");
                if (grouped) R.luts_grouped();
            }
        } else if (fcmu_batched) {
            const int npad_c = ((B * (LABEL_NC + 1) + 31) / 32) * 32;
            for (int pass = 0; pass < 2; ++pass)         // second pass: returns at once unless a projection left the f16 window
                R.check(fc_mu_batched(cd, fcmu_w_ptrs, fcmu_b_ptrs, mu_all, mu_stride, n_aces, B, npad_c, LABEL_NC + 1, SH16_ACT_SCALE,
                                      amax_slots, pass, terms == 2, st), "fc_mu (all ACEs)");
        }
        // interactive-size jobs: everything that depends on labels / codes only runs ahead on the side stream
        if (side && !prof_on && !(dbg & 4096)) {
            const bool ov = overlap_on && (long long)B * S * S > ahead_pixels && R.luts_ready;
            if (ov) {      // large job on an overlap handle: label tables of every ACE ahead on the CU-masked side stream, interior passes beside the convs
                R.prepass(lab);
                R.overlap = gtab_side != nullptr;
                R.prepare_all_ahead(lab, cd, true);
            } else if (ahead_full && (long long)B * S * S <= ahead_pixels) R.prepare_all_ahead(lab, cd, true);
            // large jobs: only the style LUT builds (small, latency-bound GEMMs) run ahead, in the tails of the conv kernels
            // (exact-f32 path, B = 16 at 512^2: 147.2 -> 148.3 images/s)
            else if ((fcmu_batched || !use_sh16) && !(dbg & 8192) && !R.luts_ready) R.prepare_all_ahead(lab, cd, false);
        }

        const int sw = S / 32;
        float* x = xa;
        float* y = xb;
        R.check(onehot_conv3x3(lab_r[5], fc_table, fc_bias, x, B, sw, sw, 16 * ngf, 0, st, use_sh16 ? 1 : 0), "fc");
        R.tap_c4("fc", x, 16 * ngf, (size_t)sw * sw);
        size_t noff = 0;
        for (const auto& b : blocks) {
            const int r = S / b.res_div;
            const size_t rr = (size_t)r * r;
            const int up = b.up_before ? 1 : 0;
            const float* xsrc = x;   // block input (at r/2 when up_before)
            const float* shortcut;
            int sc_up;
            bool fuse_s = false;
            if (b.learned) {
                R.ace(b.ace_s, lab, cd, nz, nf, noff, xsrc, up, ACT_NONE, hs);
                noff += rr;
                R.tap_sh16(b.name + ".hs", hs, b.fin, rr, b.ace_s);
                // the 1x1 shortcut conv is folded into conv_1 (extra K chunks on a second input) unless a test taps its output
                // (not on the Winograd path: there the shortcut stays a direct 1x1 GEMM whose output conv_1 adds as its residual)
                fuse_s = R.can_fuse_1x1(b.conv_1, r) && !taps.count(b.name + ".xs") && !R.use_wino(b.conv_1, r);
                if (!fuse_s) {
                    R.conv(b.conv_s, hs, b.ace_s, xs, r, nullptr, 0);
                    R.tap_c4(b.name + ".xs", xs, b.fout, rr);
                }
                shortcut = fuse_s ? nullptr : xs;
                sc_up = 0;
            } else {
                shortcut = xsrc;
                sc_up = up;
            }
            R.ace(b.ace_0, lab, cd, nz, nf, noff, xsrc, up, ACT_LRELU, h0);
            noff += rr;
            R.tap_sh16(b.name + ".h0", h0, b.fin, rr, b.ace_0);
            R.conv(b.conv_0, h0, b.ace_0, dx, r, nullptr, 0);
            R.tap_c4(b.name + ".dx", dx, b.fmid, rr);
            R.ace(b.ace_1, lab, cd, nz, nf, noff, dx, 0, ACT_LRELU, h1);
            noff += rr;
            R.tap_sh16(b.name + ".h1", h1, b.fmid, rr, b.ace_1);
            R.conv(b.conv_1, h1, b.ace_1, y, r, shortcut, sc_up, fuse_s ? &b.conv_s : nullptr, fuse_s ? hs : nullptr, fuse_s ? &b.ace_s : nullptr);
            R.tap_c4(b.name, y, b.fout, rr);
            std::swap(x, y);
        }
        R.check(conv_img_tanh(x, img_w, img_b, out + (size_t)bo * 3 * S * S, B, ngf, S, S, st, use_sh16 ? 1 : 0, img_w4), "conv_img");
        if (!R.err.empty()) return join(R.err);
    }
    return join("");
}

// phase: 0 = the whole encoder; 1 = the convolutional part only (feature map stays in the workspace; one chunk: Btot <=
// max_batch); 2 = the region means over the feature map of the preceding phase-1 call (same Btot, S).  The split lets a caller
// compute the label map concurrently with the convs: the labels are only needed by the means (architecture.py:185-205).
std::string SeanModel::encode(const float* img, const uint8_t* labels, float* codes_out, int Btot, int S,
                              hipStream_t st, int phase) {
    if (blocks.empty() || !has_zencoder) return "Zencoder weights not loaded/finalized";
    if (S % 32 != 0 || S < 32 || S > max_size) return "S must be a multiple of 32 and <= max_size";
    if (phase != 0 && Btot > max_batch) return "split encode: B must not exceed max_batch";
    if (phase == 1) { enc_pending_B = Btot; enc_pending_S = S; }
    if (phase == 2 && (enc_pending_B != Btot || enc_pending_S != S)) return "split encode: no matching ch_sean_encode_features call";
    if (phase == 2) enc_pending_B = 0;
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        std::string err;
        auto ck = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
        };
        const float* x = img + (size_t)bo * 3 * S * S;
        ConvOpts refl;
        refl.pad_mode = PAD_REFLECT;
        ConvOpts zero;
        ConvOpts zins;
        zins.in_mode = IN_UP2_ZEROINS;
        ConvOpts last = refl;
        last.act = ACT_TANH;
        const int h2 = S / 2, h4 = S / 4;
        const bool f16path = use_sh16 && h4 * h4 >= 4096;
        bool in4_done = false;       // the ConvTranspose route already applied the InstanceNorm + lrelu that follows it
        if (phase != 2) {
        ck(conv3x3_c3_reflect(x, z1_w, z1.bias, hs, B, 32, S, S, st), "zenc conv1");
        if (f16path) {
            // every conv after the stem on the f16x3 kernels: InstanceNorm + lrelu -> SH16 -> stride-2 conv (space-to-depth
            // form) -> C4, twice; then the ConvTranspose (f16x3 conv over the zero-inserted view) and the reflection-padded
            // 3x3 conv with tanh.  Instance-norm outputs are bounded by sqrt(HW): static scales (instnorm_sh16_scale).
            auto s2d = [&](const ConvLayer& L, const float* in, float* out, int hin, const char* what) {
                ConvParams q{};
                q.in = in;
                q.wpk = L.sh_wpk;
                q.wscale = L.sh_wscale;
                q.in_scale_inv = 1.f / instnorm_sh16_scale(hin * hin);
                q.out = out;
                q.B = B;
                q.Cin = L.Cin;
                q.s2d_cr = L.s2d_cr;
                q.s2d_phase0 = L.s2d_phase0;
                q.H = hin / 2;
                q.W = hin / 2;
                q.Mrows = L.Cout;
                q.bias = L.bias;
                q.act = ACT_NONE;
                ck(conv_sh16_s2d(q, L.KS, st), what);
            };
            ck(instnorm_act(hs, B * 32, S * S, 1e-5f, ACT_LRELU, st, h0, 32, splitk_ws), "zenc in1");
            s2d(z4_s2d, h0, dx, S, "zenc conv2 (f16x3)");
            ck(instnorm_c4_to_sh16(dx, B, 64, h2 * h2, 1e-5f, ACT_LRELU, h1, st, splitk_ws), "zenc in2");
            s2d(z7_s2d, h1, hs, h2, "zenc conv3 (f16x3)");
            ck(instnorm_c4_to_sh16(hs, B, 128, h4 * h4, 1e-5f, ACT_LRELU, dx, st, splitk_ws), "zenc in3");
        } else {
            ck(instnorm_act(hs, B * 32, S * S, 1e-5f, ACT_LRELU, st), "zenc in1");
            ck(run_conv(z4, hs, dx, B, S, S, zero, st), "zenc conv2");
            ck(instnorm_act(dx, B * 64, h2 * h2, 1e-5f, ACT_LRELU, st), "zenc in2");
            ck(run_conv(z7, dx, h1, B, h2, h2, zero, st), "zenc conv3");
        }
        if (f16path) {
            ConvParams t{};
            t.in = dx;
            t.out = hs;
            t.B = B;
            t.Cin = 128;
            t.bias = z10.bias;
            t.in_scale_inv = 1.f / instnorm_sh16_scale(h4 * h4);
            t.act = ACT_NONE;
            if (dbg & 16384) {           // the earlier form, kept for comparison: 3x3 conv over the zero-inserted x2 view
                t.wpk = z10_sh;
                t.wscale = z10_ws;
                t.H = h2;
                t.W = h2;
                t.Mrows = 256;
                t.in_mode = IN_UP2_ZEROINS;
                ck(conv_sh16_plain(t, 3, st), "zenc convT (f16x3)");
            } else {                     // 2x2-tap conv at the input resolution, depth-to-space store
                t.wpk = z10_d2s;
                t.wscale = z10_d2s_ws;
                t.H = h4;
                t.W = h4;
                t.Mrows = 1024;
                ck(conv_sh16_d2s(t, st), "zenc convT (f16x3, depth-to-space)");
            }
            ck(instnorm_c4_to_sh16(hs, B, 256, h2 * h2, 1e-5f, ACT_LRELU, h1, st, splitk_ws), "zenc in4");
        } else {
            ck(instnorm_act(h1, B * 128, h4 * h4, 1e-5f, ACT_LRELU, st), "zenc in3");
            const long long hw4 = (long long)h4 * h4;
            if (z10_pw[3] && !use_sh16 && !batch_inv && pw_supported(128, 256, (int)hw4) && (h4 & 3) == 0 && (long long)B * hw4 >= 16384 && vbuf &&
                (size_t)B * (512 + 4 * 256) * hw4 * sizeof(float) <= vbuf_bytes) {
                // four phase GEMMs over the shifted views (enough pixel tiles to fill the device; the V buffer is free until conv5's transform pass);
                // the InstanceNorm + lrelu that follows reads the phase planes and writes the depth-to-space plane
                float *xs = vbuf, *tph = vbuf + (size_t)B * 512 * hw4;
                ck(convt_shift4(h1, xs, B, 128, h4, h4, st), "zenc convT shifted views");
                static const int first[4] = {1, 0, 1, 0}, cnt[4] = {1, 2, 2, 4};
                for (int ph = 0; ph < 4; ++ph) {
                    PwParams q{};
                    q.in = xs + (size_t)first[ph] * 128 * hw4;
                    q.in_bs = 512 * hw4;
                    q.wpk = z10_pw[ph];
                    q.out = tph + (size_t)ph * B * 256 * hw4;
                    q.B = B;
                    q.Cin = cnt[ph] * 128;
                    q.Cout = 256;
                    q.HW = (int)hw4;
                    ck(conv_pw(q, st), "zenc convT (phase GEMM)");
                }
                ck(instnorm_act_d2s(tph, z10.bias, hs, B, 256, h4, h4, 1e-5f, ACT_LRELU, st), "zenc in4 (depth-to-space)");
                in4_done = true;
            } else if (z10_wino && !use_sh16 && wino_supported(h4, h4, 128)) {
                WinoParams q{};
                q.in = h1;
                q.wpk = z10_wino;
                q.out = hs;
                q.B = B;
                q.Cin = 128;
                q.Cout = 1024;
                q.H = h4;
                q.W = h4;
                q.bias = z10.bias;
                q.act = ACT_NONE;
                q.d2s = 1;
                q.zero = zero_page;
                ck(conv_wino_plain(q, st), "zenc convT (winograd phase convs)");
            } else
                ck(run_conv(z10, h1, hs, B, h4, h4, zins, st), "zenc convT");
            if (use_sh16) ck(instnorm_act(hs, B * 256, h2 * h2, 1e-5f, ACT_LRELU, st, h1, 256), "zenc in4");
        }
        if (use_sh16) {
            ConvParams p{};
            p.in = h1;
            p.wpk = z14_sh;
            p.out = h0;
            p.B = B;
            p.Cin = 256;
            p.H = h2;
            p.W = h2;
            p.Mrows = 512;
            p.bias = z14.bias;
            p.wscale = z14_ws;
            p.in_scale_inv = 1.f / instnorm_sh16_scale(h2 * h2);
            p.act = ACT_TANH;
            p.pad_mode = PAD_REFLECT;
            ck(conv_sh16_plain(p, 3, st), "zenc conv5 (f16x3)");
        } else {
            if (!in4_done) ck(instnorm_act(hs, B * 256, h2 * h2, 1e-5f, ACT_LRELU, st), "zenc in4");
            if (z14_wino4 && wino4_supported(h2, h2, 256) && (wino4_force || batch_inv || wino4_pays((long long)B * (h2 / 32) * (h2 / 32) * 16, num_cus))) {
                Wino4Params q{};
                q.in = hs;
                q.wpk = z14_wino4;
                q.out = h0;
                q.B = B;
                q.Cin = 256;
                q.Cout = 512;
                q.H = h2;
                q.W = h2;
                q.bias = z14.bias;
                q.act = ACT_TANH;
                q.reflect = 1;
                if (wino4v && wino4v_fits(B, h2, 64)) {          // 512 GEMM rows: the (reflected) input transform in its own pass
                    Wino4vPackParams vp{};
                    vp.in = hs;
                    vp.v = vbuf;
                    vp.B = B;
                    vp.K = 256;
                    vp.H = vp.W = vp.pitch = h2;
                    vp.nks = 64;
                    vp.reflect = 1;
                    ck(wino4v_pack(vp, st), "zenc conv5 input transform");
                    q.v = vbuf;
                    ck(conv_wino4v_plain(q, st), "zenc conv5 (winograd F(4x4,3x3), pre-transformed input)");
                } else
                    ck(conv_wino4_plain(q, st), "zenc conv5 (winograd F(4x4,3x3))");
            } else if (z14_wino && wino_supported(h2, h2, 256)) {
                WinoParams q{};
                q.in = hs;
                q.wpk = z14_wino;
                q.out = h0;
                q.B = B;
                q.Cin = 256;
                q.Cout = 512;
                q.H = h2;
                q.W = h2;
                q.bias = z14.bias;
                q.act = ACT_TANH;
                q.reflect = 1;
                q.zero = zero_page;
                ck(conv_wino_plain(q, st), "zenc conv5 (winograd)");
            } else
                ck(run_conv(z14, hs, h0, B, h2, h2, last, st), "zenc conv5");
        }
        }   // phase != 2
        if (phase == 1) {
            if (!err.empty()) return err;
            continue;
        }
        ck(region_mean(h0, labels + (size_t)bo * S * S, codes_out + (size_t)bo * LABEL_NC * STYLE, B, STYLE, h2, h2, S, st,
                       use_sh16 ? 1 : 0), "region_mean");
        auto it = taps.find("zenc.feat");
        if (it != taps.end() && it->second) {
            if (use_sh16) ck(c4_decode(h0, it->second, B, STYLE, (long long)h2 * h2, st), "tap");
            else ck(hipMemcpyAsync(it->second, h0, (size_t)B * STYLE * h2 * h2 * 4, hipMemcpyDeviceToDevice, st), "tap");
        }
        if (!err.empty()) return err;
    }
    return "";
}

bool SeanModel::wino4v_fits(int Bn, int r, int nks) const { return vbuf && r % 32 == 0 && wino4v_bytes(Bn, r, r, nks) <= vbuf_bytes; }

}  // namespace chk
