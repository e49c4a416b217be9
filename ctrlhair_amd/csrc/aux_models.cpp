// aux_models.cpp -- colour/texture MLPs, shape VAE, BiSeNet: weight folding + launch schedules.
// Reference behaviour restated (file:line in /root/reference) is cited at each forward.
#include "aux_models.h"

#include "conv_sh16.h"
#include "kernels.h"
#include "sh16.h"

namespace chk {

namespace {
struct Ck {
    std::string err;
    void operator()(hipError_t e, const char* what) {
        if (e != hipSuccess && err.empty()) err = std::string(what) + ": " + hipGetErrorString(e);
    }
};
void free_all(std::vector<void*>& a) {
    for (void* p : a) (void)hipFree(p);
    a.clear();
}
// eval BatchNorm -> (scale, shift)
void bn_fold(Builder& B, const std::string& p, int C, std::vector<float>& scale, std::vector<float>& shift,
             float eps = 1e-5f) {
    auto g = B.vec(p + ".weight", C), b = B.vec(p + ".bias", C), rm = B.vec(p + ".running_mean", C),
         rv = B.vec(p + ".running_var", C);
    scale.resize(C);
    shift.resize(C);
    for (int c = 0; c < C; ++c) {
        scale[c] = g[c] / std::sqrt(rv[c] + eps);
        shift[c] = b[c] - rm[c] * scale[c];
    }
}
}  // namespace

// =================================================================================================================
// Colour / texture branch
// =================================================================================================================
std::string ColorModel::build(const TensorStore& ts, int mb) {
    Builder B(ts, allocs);
    max_batch = mb;
    // EigenGenerator (model_eigengan.py:34-60): Linear 5->256, 4 subspaces (2-d), 3x Linear 256->256, Linear 256->512
    g_in_w = B.upload(B.vec("gen.main_layer_in.weight", 256 * 5));
    g_in_b = B.upload(B.vec("gen.main_layer_in.bias", 256));
    for (int k = 0; k < 4; ++k) {
        const int o = k == 3 ? 512 : 256;
        const std::string p = "gen.main_layer_mid." + std::to_string(k) + ".1";
        g_mid_w[k] = B.upload(B.vec(p + ".weight", (size_t)o * 256));
        g_mid_b[k] = B.upload(B.vec(p + ".bias", o));
        const std::string s = "gen.subspaces." + std::to_string(k);
        sub_U[k] = B.upload(B.vec(s + ".U", 2 * 256));
        sub_L[k] = B.upload(B.vec(s + ".L", 2));
        sub_mu[k] = B.upload(B.vec(s + ".mu", 256));
    }
    // Discriminator used as encoder (model.py:86-106): 512->256, 3x 256->256 (lrelu 0.2), 256->11
    for (int k = 0; k < 5; ++k) {
        const int i = k == 0 ? 512 : 256, o = k == 4 ? 11 : 256;
        const std::string p = "dis.net." + std::to_string(k) + ".fc";
        d_w[k] = B.upload(B.vec(p + ".weight", (size_t)o * i));
        d_b[k] = B.upload(B.vec(p + ".bias", o));
    }
    // Predictor p004 (predictor_model.py:14-30): 3x (Linear + BatchNorm1d(eval) + lrelu), Linear 256->4
    for (int k = 0; k < 4; ++k) {
        const int i = k == 0 ? 512 : 256, o = k == 3 ? 4 : 256;
        const std::string p = "rgb.net." + std::to_string(k);
        p_w[k] = B.upload(B.vec(p + ".fc.weight", (size_t)o * i));
        p_b[k] = B.upload(B.vec(p + ".fc.bias", o));
        if (k < 3) {
            std::vector<float> sc, sh;
            bn_fold(B, p + ".norm", 256, sc, sh);
            p_scale[k] = B.upload(sc);
            p_shift[k] = B.upload(sh);
        }
    }
    wa = B.falloc((size_t)mb * 512);
    wb = B.falloc((size_t)mb * 512);
    if (!B.err.empty()) return B.err;
    if (hipDeviceSynchronize() != hipSuccess) return "device sync failed";
    ready = true;
    return "";
}
void ColorModel::destroy() { free_all(allocs); ready = false; }

// model_eigengan.py:62-84: x = Linear([curl, rgb, pca_std]); 4x { x += (L*z_k)U_k + mu_k ; x = Linear(lrelu(x)) }
std::string ColorModel::generate(const float* noise, const float* cond, float* code, int Btot, hipStream_t st) {
    if (!ready) return "colour model not finalized";
    Ck ck;
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        ck(linear(cond + (size_t)bo * 5, g_in_w, g_in_b, nullptr, nullptr, wa, B, 5, 256, 5, 256, ACT_NONE, st), "gen in");
        float *x = wa, *y = wb;
        for (int k = 0; k < 4; ++k) {
            ck(subspace_add(x, noise + (size_t)bo * 8 + 2 * k, 8, sub_U[k], sub_L[k], sub_mu[k], B, 256, 2, st), "subspace");
            const bool last = k == 3;
            ck(linear(x, g_mid_w[k], g_mid_b[k], nullptr, nullptr, last ? code + (size_t)bo * 512 : y, B, 256,
                      last ? 512 : 256, 256, last ? 512 : 256, ACT_NONE, st), "gen mid");
            std::swap(x, y);
        }
    }
    return ck.err;
}
// model.py:108-111 (self.net): returns the raw 11-vector; the host shim slices adv/noise/noise_curliness (:112-127)
std::string ColorModel::encode(const float* code, float* out11, int Btot, hipStream_t st) {
    if (!ready) return "colour model not finalized";
    Ck ck;
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        const float* x = code + (size_t)bo * 512;
        float *a = wa, *b = wb;
        for (int k = 0; k < 5; ++k) {
            const int i = k == 0 ? 512 : 256, o = k == 4 ? 11 : 256;
            float* dst = k == 4 ? out11 + (size_t)bo * 11 : a;
            ck(linear(x, d_w[k], d_b[k], nullptr, nullptr, dst, B, i, o, i, o, k == 4 ? ACT_NONE : ACT_LRELU, st), "dis");
            x = a;
            std::swap(a, b);
        }
    }
    return ck.err;
}
// predictor_model.py:32-41
std::string ColorModel::predict(const float* code, float* out4, int Btot, hipStream_t st) {
    if (!ready) return "colour model not finalized";
    Ck ck;
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        const float* x = code + (size_t)bo * 512;
        float *a = wa, *b = wb;
        for (int k = 0; k < 4; ++k) {
            const int i = k == 0 ? 512 : 256, o = k == 3 ? 4 : 256;
            float* dst = k == 3 ? out4 + (size_t)bo * 4 : a;
            ck(linear(x, p_w[k], p_b[k], k < 3 ? p_scale[k] : nullptr, k < 3 ? p_shift[k] : nullptr, dst, B, i, o, i, o,
                      k == 3 ? ACT_NONE : ACT_LRELU, st), "predictor");
            x = a;
            std::swap(a, b);
        }
    }
    return ck.err;
}

// =================================================================================================================
// Shape branch
// =================================================================================================================
std::string ShapeModel::build(const TensorStore& ts, int mb) {
    Builder B(ts, allocs);
    max_batch = mb;
    const char* side[2] = {"hair", "face"};
    for (int w = 0; w < 2; ++w) {
        // MaskEncoder (shape_branch/model.py:69-94): 7x Conv2dBlock(k4, s2, ZeroPad 1, LayerNorm, lrelu)
        int cin = (w == 0 ? 1 : 18) + 40;
        for (int l = 0; l < 7; ++l) {
            const int cout = std::min(2048, 32 << l);
            const std::string p = std::string(side[w]) + "_encoder.layers." + std::to_string(l);
            const auto gam = B.vec(p + ".norm.gamma", cout), bet = B.vec(p + ".norm.beta", cout);
            if (use_sh16 && l < ENC_S2D) {
                enc_s2d[w][l] = make_conv_s2d(B, B.vec(p + ".conv.weight", (size_t)cout * cin * 16), B.vec(p + ".conv.bias", cout),
                                              cout, cin, 4);
                float gm = 0.f, bm = 0.f;
                for (int c = 0; c < cout; ++c) { gm = std::max(gm, std::fabs(gam[c])); bm = std::max(bm, std::fabs(bet[c])); }
                const int sz = S >> (l + 1);          // output side of layer l; |LN output| <= sqrt(N) max|gamma| + max|beta|
                enc_ln_scale[w][l] = sh16_scale_for_bound(std::sqrt((float)cout * sz * sz) * gm + bm);
            } else {
                const auto wv0 = B.vec(p + ".conv.weight", (size_t)cout * cin * 16);
                enc[w][l] = make_conv(B, wv0, B.vec(p + ".conv.bias", cout), cout, cin, 4, 2, 1);
                if (l == 0 && !use_sh16 && enc_l0_lut && wv0.size() == (size_t)cout * cin * 16) {
                    // the mask channels' weights as table rows [tap][label][channel] (misc_kernels.hip shape_enc_l0): hair = class 13 -> input
                    // channel 0; face = class l != 13 -> input channel l (l < 13) or l - 1 (shape_util.py:23-26); row 19 and unused rows: zeros
                    if (enc0_tab_host.empty()) enc0_tab_host.assign((size_t)2 * 16 * 20 * 32, 0.f);
                    for (int t = 0; t < 16; ++t)
                        for (int lab = 0; lab < 19; ++lab) {
                            const int ch = w == 0 ? (lab == 13 ? 0 : -1) : (lab == 13 ? -1 : (lab < 13 ? lab : lab - 1));
                            if (ch < 0) continue;
                            for (int c = 0; c < 32; ++c)
                                enc0_tab_host[(((size_t)w * 16 + t) * 20 + lab) * 32 + c] = wv0[((size_t)c * cin + ch) * 16 + t];
                        }
                }
            }
            enc_ln[w][l].gamma = B.upload(gam);
            enc_ln[w][l].beta = B.upload(bet);
            cin = cout;
        }
        const int odim = w == 0 ? HAIR_DIM : FACE_DIM;
        const std::string q = std::string(side[w]) + "_encoder.out_layer.fc";
        enc_fc_w[w] = B.upload(B.vec(q + ".weight", (size_t)odim * 8192));
        enc_fc_b[w] = B.upload(B.vec(q + ".bias", odim));
        // MaskDecoder (:116-136): Linear -> [2048,2,2]; 7x (nearest x2, Conv2dBlock k3 p1 LN lrelu); Conv2dBlock k3 -> 1 / 18
        const int idim = w == 0 ? FACE_DIM + HAIR_DIM : FACE_DIM;
        const std::string d = std::string(side[w]) + "_decoder";
        dec_in_w[w] = B.upload(B.vec(d + ".in_layer.fc.weight", (size_t)8192 * idim));
        dec_in_b[w] = B.upload(B.vec(d + ".in_layer.fc.bias", 8192));
        int ci = 2048;
        for (int l = 0; l < 7; ++l) {
            const int co = std::min(32 << (6 - l), 2048);
            const std::string p = d + ".layers." + std::to_string(2 * l + 1);
            const auto wv = B.vec(p + ".conv.weight", (size_t)co * ci * 9);
            const auto gam = B.vec(p + ".norm.gamma", co), bet = B.vec(p + ".norm.beta", co);
            if (use_sh16 && l >= 1) {          // f16x3 path: only the bias of the f32 layer object is used
                const float* wp = wv.data();
                auto getw = [&](int row, int c, int t) { return wp[((size_t)row * ci + c) * 9 + t]; };
                const auto kexp = sh16_row_exponents(co, ci, 3, getw);
                dec_sh[w][l] = B.upload(pack_A_sh16(co, ci, 3, getw, kexp));
                dec_ws[w][l] = B.upload(sh16_wscale(kexp));
                dec[w][l].bias = B.upload(B.vec(p + ".conv.bias", co));
                dec[w][l].Cout = co;
                dec[w][l].Cin = ci;
                dec[w][l].KS = 3;
            } else {
                dec[w][l] = make_conv(B, wv, B.vec(p + ".conv.bias", co), co, ci, 3, 1, 1, wino && (4 << l) >= 16);      // (output side 4 << l)
            }
            dec_ln[w][l].gamma = B.upload(gam);
            dec_ln[w][l].beta = B.upload(bet);
            {   // |(x - mean) / (std + eps)| <= sqrt(N) over the N = C*H*W elements of a sample
                float gm = 0.f, bm = 0.f;
                for (int c = 0; c < co; ++c) { gm = std::max(gm, std::fabs(gam[c])); bm = std::max(bm, std::fabs(bet[c])); }
                const int sz = 4 << l;            // output side of layer l
                dec_ln_scale[w][l] = sh16_scale_for_bound(std::sqrt((float)co * sz * sz) * gm + bm);
            }
            ci = co;
        }
        const int oc = w == 0 ? 1 : 18;
        if (use_sh16)
            dec_out_sh[w] = make_conv_sh16(B, B.vec(d + ".out_layer.conv.weight", (size_t)oc * 32 * 9),
                                           B.vec(d + ".out_layer.conv.bias", oc), oc, 32, 3, 1);
        else
            dec_out[w] = make_conv(B, B.vec(d + ".out_layer.conv.weight", (size_t)oc * 32 * 9), B.vec(d + ".out_layer.conv.bias", oc),
                                   oc, 32, 3, 1, 1);
        if (!B.err.empty()) return B.err;
    }
    // generate_pos_embedding (shape_branch/model.py:18-30): sin/cos(2^k pi u) for u in {x, y} grids, order 10.
    // channel layout after reshape([-1,S,S]): [sin: k0(x,y), k1(x,y) ...][cos: ...]  (gamma1 then gamma2, each [order,2,S,S])
    {
        std::vector<float> pe((size_t)40 * S * S);
        for (int part = 0; part < 2; ++part)
            for (int k = 0; k < 10; ++k)
                for (int ax = 0; ax < 2; ++ax) {
                    const int ch = part * 20 + k * 2 + ax;
                    const double f = std::ldexp(1.0, k) * 3.14159265358979323846;
                    for (int y = 0; y < S; ++y)
                        for (int x = 0; x < S; ++x) {
                            // np.meshgrid(c, c) default 'xy': [0] varies along columns (x), [1] along rows (y)
                            const double u = (ax == 0 ? x : y) / (double)S;
                            pe[((size_t)ch * S + y) * S + x] = (float)(part == 0 ? std::sin(f * u) : std::cos(f * u));
                        }
                }
        pos = B.upload(pe);
    }
    const size_t HW = (size_t)S * S;
    in_hair = B.falloc(mb * 48 * HW);      // f16x3 path: SH16 with the channels padded to 48 / 64 (4 bytes per element either way)
    in_face = B.falloc(mb * 64 * HW);
    bufa = B.falloc((size_t)mb * (8192 + 24 * HW));   // decoder input vector + temporary logits (f16x3 path: C4, 4 + 20 rows)
    bufb = B.falloc(mb * 32 * HW);                      // ping-pong activations (largest: 32 ch at 256^2)
    bufc = B.falloc(mb * 32 * HW);
    lnpart = B.falloc((size_t)mb * 128 * 3);
    splitk_cap = (long long)8 << 20;
    splitk_ws = B.falloc((size_t)splitk_cap);
    codecat = B.falloc((size_t)mb * (FACE_DIM + HAIR_DIM));
    // second workspace set + side stream: the hair encoder / decoder beside the face one (option "shape.overlap")
    bufa2 = bufb2 = bufc2 = lnpart2 = splitk_ws2 = nullptr;
    if (overlap) {
        bufa2 = B.falloc((size_t)mb * 8192);
        bufb2 = B.falloc(mb * 32 * HW);
        bufc2 = B.falloc(mb * 32 * HW);
        lnpart2 = B.falloc((size_t)mb * 128 * 3);
        splitk_ws2 = B.falloc((size_t)splitk_cap);
        if (!side_stream && (hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess))
            return "shape model: side stream / events";
    }
    enc0_tab = nullptr;
    enc0_pc[0] = enc0_pc[1] = nullptr;
    if (!use_sh16 && enc_l0_lut && !enc0_tab_host.empty() && B.err.empty()) {
        // layer 0 of the encoders as a label table: the positional channels' part of the conv (+ bias) comes from the conv kernel itself,
        // run once on a label map without any class (every one-hot channel zero)
        enc0_tab = B.upload(enc0_tab_host);
        std::vector<float> none((size_t)(HW + 3) / 4);
        std::memset(none.data(), 0xFF, none.size() * sizeof(float));
        const uint8_t* lab_none = reinterpret_cast<const uint8_t*>(B.upload(none));
        enc0_pc[0] = B.falloc((size_t)32 * HW / 4);
        enc0_pc[1] = B.falloc((size_t)32 * HW / 4);
        if (B.err.empty()) {
            Ck ck;
            ck(shape_inputs(lab_none, pos, in_hair, in_face, 1, (int)HW, nullptr), "shape encoder layer 0: inputs of the positional part");
            ck(run_conv(enc[0][0], in_hair, enc0_pc[0], 1, S, S, ConvOpts(), nullptr), "shape encoder layer 0: positional part (hair)");
            ck(run_conv(enc[1][0], in_face, enc0_pc[1], 1, S, S, ConvOpts(), nullptr), "shape encoder layer 0: positional part (face)");
            if (!ck.err.empty()) return ck.err;
        }
    }
    enc0_tab_host.clear();
    enc0_tab_host.shrink_to_fit();
    if (!B.err.empty()) return B.err;
    if (hipDeviceSynchronize() != hipSuccess) return "device sync failed";
    ready = true;
    return "";
}
void ShapeModel::destroy() {
    free_all(allocs);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side_stream) (void)hipStreamDestroy(side_stream);
    ev_fork = ev_join = nullptr;
    side_stream = nullptr;
    ready = false;
}

// MaskEncoder.forward (shape_branch/model.py:96-108); vae std head unused at test (:164-169 testing=True)
std::string ShapeModel::run_encoder(int w, const float* in, float* code, int B, hipStream_t st, int set) {
    Ck ck;
    // workspace set: 0 = the caller's stream, 1 = the side stream (the two encoders are independent and each one alone leaves most of
    // the chip idle: encode() runs them side by side)
    float *wb = set ? bufb2 : bufb, *wc = set ? bufc2 : bufc, *wl = set ? lnpart2 : lnpart, *wk = set ? splitk_ws2 : splitk_ws;
    const float* x = in;
    float* bufs[2] = {wb, wc};
    int size = S;
    int l0 = 0;
    if (use_sh16) {
        // layers 0..ENC_S2D-1: SH16 -> f16x3 stride-2 conv (space-to-depth form) -> C4 -> LayerNorm + lrelu -> SH16 (last: NCHW f32)
        float s_in = ENC_IN_SCALE;
        for (int l = 0; l < ENC_S2D; ++l) {
            const ConvLayer& L = enc_s2d[w][l];
            ConvParams p{};
            p.in = x;
            p.wpk = L.sh_wpk;
            p.wscale = L.sh_wscale;
            p.in_scale_inv = 1.f / s_in;
            p.out = wc;
            p.B = B;
            p.Cin = L.Cin;
            p.s2d_cr = L.s2d_cr;
            p.s2d_phase0 = L.s2d_phase0;
            size /= 2;
            p.H = size;
            p.W = size;
            p.Mrows = L.Cout;
            p.bias = L.bias;
            p.act = ACT_NONE;
            p.partial = wk;
            p.partial_cap = splitk_cap;
            ck(conv_sh16_s2d(p, L.KS, st), "shape enc conv (f16x3)");
            ck(layernorm_act_conv(wc, 1, wb, l < ENC_S2D - 1 ? 1 : 0, enc_ln_scale[w][l], enc_ln[w][l].gamma, enc_ln[w][l].beta, wl, B,
                                  L.Cout, size * size, 1e-5f, ACT_LRELU, st), "shape enc ln");
            x = wb;
            s_in = enc_ln_scale[w][l];
        }
        l0 = ENC_S2D;
        bufs[0] = wc;              // layer 4 reads wb
        bufs[1] = wb;
    }
    for (int l = l0; l < 7; ++l) {
        float* y = bufs[l & 1];
        ConvOpts eo;
        eo.partial = wk;
        eo.partial_cap = splitk_cap;
        if (l == 0 && enc0_tab) {
            // `in` already holds layer 0's conv output (encode(): the label-table kernel wrote both encoders'): LayerNorm in place
            y = const_cast<float*>(in);
        } else {
            ck(run_conv(enc[w][l], x, y, B, size, size, eo, st), "shape enc conv");
        }
        size /= 2;
        ck(layernorm_act(y, enc_ln[w][l].gamma, enc_ln[w][l].beta, wl, B, enc[w][l].Cout, size * size, 1e-5f, ACT_LRELU, st),
           "shape enc ln");
        x = y;
    }
    const int odim = w == 0 ? HAIR_DIM : FACE_DIM;
    ck(linear(x, enc_fc_w[w], enc_fc_b[w], nullptr, nullptr, code, B, 8192, odim, 8192, odim, ACT_NONE, st), "shape enc fc");
    return ck.err;
}

// ui/backend.py:81-86: one-hot, split_hair_face, forward_hair_encoder(testing=True), forward_face_encoder
std::string ShapeModel::encode(const uint8_t* labels, float* hair_code, float* face_code, int Btot, hipStream_t st) {
    if (!ready) return "shape model not finalized";
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        Ck ck;
        if (use_sh16) ck(shape_inputs_sh16(labels + (size_t)bo * S * S, pos, in_hair, in_face, B, S * S, ENC_IN_SCALE, st), "shape inputs");
        else if (enc0_tab)       // exact f32: no one-hot / positional input tensors at all -- layer 0 of both encoders straight from the labels
            ck(shape_enc_l0(labels + (size_t)bo * S * S, enc0_tab, enc0_pc[0], enc0_pc[1], hair_code ? in_hair : nullptr, face_code ? in_face : nullptr,
                            B, S, st), "shape encoder layer 0 (label table)");
        else ck(shape_inputs(labels + (size_t)bo * S * S, pos, in_hair, in_face, B, S * S, st), "shape inputs");
        if (!ck.err.empty()) return ck.err;
        std::string e;
        const bool par = overlap && side_stream && hair_code && face_code;      // the hair encoder on the side stream, beside the face encoder
        if (par) {
            ck(hipEventRecord(ev_fork, st), "shape encode fork");
            ck(hipStreamWaitEvent(side_stream, ev_fork, 0), "shape encode fork wait");
            if (!ck.err.empty()) return ck.err;
        }
        if (hair_code) e = run_encoder(0, in_hair, hair_code + (size_t)bo * HAIR_DIM, B, par ? side_stream : st, par ? 1 : 0);
        if (!e.empty()) return e;
        if (par) ck(hipEventRecord(ev_join, side_stream), "shape encode join");
        if (face_code) e = run_encoder(1, in_face, face_code + (size_t)bo * FACE_DIM, B, st, 0);
        if (!e.empty()) return e;
        if (par) ck(hipStreamWaitEvent(st, ev_join, 0), "shape encode join wait");
        if (!ck.err.empty()) return ck.err;
    }
    return "";
}

// MaskDecoder.forward (shape_branch/model.py:138-143)
std::string ShapeModel::run_decoder(int w, const float* code, int code_dim, float* logit, int B, hipStream_t st, int set) {
    Ck ck;
    float *wa = set ? bufa2 : bufa, *wb = set ? bufb2 : bufb, *wc = set ? bufc2 : bufc, *wl = set ? lnpart2 : lnpart, *wk = set ? splitk_ws2 : splitk_ws;
    ck(linear(code, dec_in_w[w], dec_in_b[w], nullptr, nullptr, wa, B, code_dim, 8192, code_dim, 8192, ACT_NONE, st), "dec in");
    const float* x = wa;   // [B,2048,2,2]
    float* bufs[2] = {wb, wc};
    int size = 2;
    ConvOpts up;
    up.in_mode = IN_UP2_NEAREST;
    up.partial = wk;
    up.partial_cap = splitk_cap;
    up.no_wino = !wino;
    ConvOpts plain_o;
    plain_o.no_wino = !wino;
    if (use_sh16) {
        // layer 0 (2x2 -> 4x4, input straight from the Linear) on the exact-f32 kernel; its LayerNorm writes SH16.  Layers 1-6:
        // f16x3 conv over the nearest-x2 view (SH16 in, C4 out) -> LayerNorm + lrelu (C4 in, SH16 out); output conv -> C4 logits.
        ck(run_conv(dec[w][0], x, wc, B, size, size, up, st), "shape dec conv0");
        size *= 2;
        ck(layernorm_act_conv(wc, 0, wb, 1, dec_ln_scale[w][0], dec_ln[w][0].gamma, dec_ln[w][0].beta, wl, B,
                              dec[w][0].Cout, size * size, 1e-5f, ACT_LRELU, st), "shape dec ln0");
        for (int l = 1; l < 7; ++l) {
            ConvParams p{};
            p.in = wb;
            p.wpk = dec_sh[w][l];
            p.wscale = dec_ws[w][l];
            p.in_scale_inv = 1.f / dec_ln_scale[w][l - 1];
            p.out = wc;
            p.B = B;
            p.Cin = dec[w][l].Cin;
            p.H = 2 * size;
            p.W = 2 * size;
            p.Mrows = dec[w][l].Cout;
            p.bias = dec[w][l].bias;
            p.act = ACT_NONE;
            p.in_mode = IN_UP2_NEAREST;
            p.partial = wk;
            p.partial_cap = splitk_cap;
            ck(conv_sh16_plain(p, 3, st), "shape dec conv (f16x3)");
            size *= 2;
            ck(layernorm_act_conv(wc, 1, wb, 1, dec_ln_scale[w][l], dec_ln[w][l].gamma, dec_ln[w][l].beta, wl, B,
                                  dec[w][l].Cout, size * size, 1e-5f, ACT_LRELU, st), "shape dec ln");
        }
        {   // output conv (32 -> 1 / 18 rows, padded): C4 logits
            const ConvLayer& L = dec_out_sh[w];
            ConvParams p{};
            p.in = wb;
            p.wpk = L.sh_wpk;
            p.wscale = L.sh_wscale;
            p.in_scale_inv = 1.f / dec_ln_scale[w][6];
            p.out = logit;
            p.B = B;
            p.Cin = 32;
            p.H = size;
            p.W = size;
            p.Mrows = L.Cout;
            p.bias = L.bias;
            p.act = ACT_NONE;
            ck(conv_sh16_plain(p, 3, st), "shape dec out (f16x3)");
        }
        return ck.err;
    }
    for (int l = 0; l < 7; ++l) {
        float* y = bufs[l & 1];
        ck(run_conv(dec[w][l], x, y, B, size, size, up, st), "shape dec conv");
        size *= 2;
        ck(layernorm_act(y, dec_ln[w][l].gamma, dec_ln[w][l].beta, wl, B, dec[w][l].Cout, size * size, 1e-5f, ACT_LRELU, st),
           "shape dec ln");
        x = y;
    }
    ck(run_conv(dec_out[w], x, logit, B, size, size, plain_o, st), "shape dec out");
    return ck.err;
}

// Generator.forward_decoder (shape_branch/model.py:184-187) + mask_one_hot_to_label (shape_util.py:17-20)
std::string ShapeModel::combine(const float* hair_logit, const float* face_logit, uint8_t* labels, float* probs, int B,
                                hipStream_t st) {
    if (!ready) return "shape model not finalized";
    Ck ck;
    ck(shape_softmax(hair_logit, face_logit, labels, probs, B, S * S, st), "shape softmax");
    return ck.err;
}

// forward_decode_by_code (:195-199) = forward_hair_decoder (cat[face_code, hair_code], :175-178) + forward_face_decoder
std::string ShapeModel::decode(const float* hair_code, const float* face_code, float* hair_logit, float* face_logit,
                               uint8_t* labels, float* probs, int Btot, hipStream_t st) {
    if (!ready) return "shape model not finalized";
    const size_t HW = (size_t)S * S;
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        // logits: caller buffers when given, else the tail of bufa (the decoder only uses bufa's first B*8192 floats).
        // f16x3 path: the decoders write C4 logits (4 / 20 padded rows) into that tail; callers' NCHW buffers are filled from it
        float* hl = hair_logit && !use_sh16 ? hair_logit + bo * HW : bufa + (size_t)B * 8192;
        float* fl = face_logit && !use_sh16 ? face_logit + bo * 18 * HW : bufa + (size_t)B * 8192 + (use_sh16 ? 4 : 1) * B * HW;
        std::string e;
        const bool par = overlap && side_stream && hair_code;       // the hair decoder on the side stream, beside the face decoder
        if (hair_code) {
            Ck ck;
            hipStream_t hs = par ? side_stream : st;
            if (par) {
                ck(hipEventRecord(ev_fork, st), "shape decode fork");
                ck(hipStreamWaitEvent(side_stream, ev_fork, 0), "shape decode fork wait");
            }
            ck(hipMemcpy2DAsync(codecat, (FACE_DIM + HAIR_DIM) * 4, face_code + (size_t)bo * FACE_DIM, FACE_DIM * 4, FACE_DIM * 4, B,
                                hipMemcpyDeviceToDevice, hs), "cat face");
            ck(hipMemcpy2DAsync(codecat + FACE_DIM, (FACE_DIM + HAIR_DIM) * 4, hair_code + (size_t)bo * HAIR_DIM, HAIR_DIM * 4,
                                HAIR_DIM * 4, B, hipMemcpyDeviceToDevice, hs), "cat hair");
            if (!ck.err.empty()) return ck.err;
            e = run_decoder(0, codecat, FACE_DIM + HAIR_DIM, hl, B, hs, par ? 1 : 0);
            if (!e.empty()) return e;
            if (par) ck(hipEventRecord(ev_join, side_stream), "shape decode join");
            if (!ck.err.empty()) return ck.err;
        }
        e = run_decoder(1, face_code + (size_t)bo * FACE_DIM, FACE_DIM, fl, B, st, 0);
        if (!e.empty()) return e;
        if (par && hipStreamWaitEvent(st, ev_join, 0) != hipSuccess) return "shape decode join wait failed";
        if (use_sh16) {
            Ck ck;
            if (hair_code && hair_logit) ck(c4_rows_to_nchw(hl, hair_logit + bo * HW, B, 1, 4, (int)HW, st), "hair logits");
            if (face_logit) ck(c4_rows_to_nchw(fl, face_logit + bo * 18 * HW, B, 18, 20, (int)HW, st), "face logits");
            if (labels && hair_code)
                ck(shape_softmax(hl, fl, labels + bo * HW, probs ? probs + bo * 19 * HW : nullptr, B, (int)HW, st, 1), "shape softmax");
            if (!ck.err.empty()) return ck.err;
        } else if (labels && hair_code) {
            e = combine(hl, fl, labels + bo * HW, probs ? probs + bo * 19 * HW : nullptr, B, st);
            if (!e.empty()) return e;
        }
    }
    return "";
}

// =================================================================================================================
// BiSeNet
// =================================================================================================================
namespace {
// conv (no bias) + eval BN folded: w' = w * s[co], b' = shift[co]
// sh16: packed for the f16x3 kernels instead (stride 2: in the space-to-depth form)
ConvLayer conv_bn(Builder& B, const std::string& conv, const std::string& bn, int cout, int cin, int ks, int stride,
                  int pad, bool sh16 = false, bool want_wino = true) {
    auto w = B.vec(conv + ".weight", (size_t)cout * cin * ks * ks);
    std::vector<float> sc, sh;
    bn_fold(B, bn, cout, sc, sh);
    for (int o = 0; o < cout; ++o)
        for (size_t i = 0; i < (size_t)cin * ks * ks; ++i) w[(size_t)o * cin * ks * ks + i] *= sc[o];
    if (sh16 && stride == 1) return make_conv_sh16(B, w, sh, cout, cin, ks, pad);
    if (sh16) return make_conv_s2d(B, w, sh, cout, cin, ks);        // stride 2: space-to-depth form
    return make_conv(B, w, sh, cout, cin, ks, stride, pad, want_wino);
}
}  // namespace

std::string BiSeNetModel::build(const TensorStore& ts, int mb, int ms) {
    Builder B(ts, allocs);
    if (ms % 32 != 0 || ms < 64) return "BiSeNet max_size must be a multiple of 32 (>= 64)";
    max_batch = mb;
    max_size = ms;
    {   // stem (resnet.py:61-62): 7x7 s2 conv + BN folded, VALU kernel layout [3*49][64]
        auto w = B.vec("cp.resnet.conv1.weight", 64 * 147);
        std::vector<float> sc, sh;
        bn_fold(B, "cp.resnet.bn1", 64, sc, sh);
        std::vector<float> wt((size_t)147 * 64);            // [tap = (c, ky, kx)][output channel]: stem7x7_kernel's scalar loads
        for (int o = 0; o < 64; ++o)
            for (int i = 0; i < 147; ++i) wt[(size_t)i * 64 + o] = w[o * 147 + i] * sc[o];
        stem_w = B.upload(wt);
        stem_b = B.upload(sh);
    }
    const int chans[5] = {64, 64, 128, 256, 512};
    for (int L = 1; L <= 4; ++L)
        for (int i = 0; i < 2; ++i) {
            BasicBlockW& bb = blk[(L - 1) * 2 + i];
            const int cin = i == 0 ? chans[L - 1] : chans[L], cout = chans[L];
            const int stride = (i == 0 && L > 1) ? 2 : 1;
            const std::string p = "cp.resnet.layer" + std::to_string(L) + "." + std::to_string(i);
            bb.c1 = conv_bn(B, p + ".conv1", p + ".bn1", cout, cin, 3, stride, 1, use_sh16, wino != 0);
            bb.c2 = conv_bn(B, p + ".conv2", p + ".bn2", cout, cout, 3, 1, 1, use_sh16, wino != 0);
            bb.has_down = (cin != cout || stride != 1);
            if (bb.has_down) bb.down = conv_bn(B, p + ".downsample.0", p + ".downsample.1", cout, cin, 1, stride, 0, use_sh16, wino != 0);
        }
    arm16_conv = conv_bn(B, "cp.arm16.conv.conv", "cp.arm16.conv.bn", 128, 256, 3, 1, 1, use_sh16, wino != 0);
    arm32_conv = conv_bn(B, "cp.arm32.conv.conv", "cp.arm32.conv.bn", 128, 512, 3, 1, 1, use_sh16, wino != 0);
    head32 = conv_bn(B, "cp.conv_head32.conv", "cp.conv_head32.bn", 128, 128, 3, 1, 1, use_sh16, wino != 0);
    head16 = conv_bn(B, "cp.conv_head16.conv", "cp.conv_head16.bn", 128, 128, 3, 1, 1, use_sh16, wino != 0);
    {   // FFM convblk (1x1, 256 -> 256 on cat[fsp, fcp]) split into the two 128-channel halves (no concat buffer)
        auto w = B.vec("ffm.convblk.conv.weight", 256 * 256);
        std::vector<float> sc, sh;
        bn_fold(B, "ffm.convblk.bn", 256, sc, sh);
        std::vector<float> wa((size_t)256 * 128), wb((size_t)256 * 128);
        for (int o = 0; o < 256; ++o)
            for (int i = 0; i < 128; ++i) {
                wa[(size_t)o * 128 + i] = w[(size_t)o * 256 + i] * sc[o];
                wb[(size_t)o * 128 + i] = w[(size_t)o * 256 + 128 + i] * sc[o];
            }
        ffm_a = use_sh16 ? make_conv_sh16(B, wa, std::vector<float>(), 256, 128, 1, 0) : make_conv(B, wa, std::vector<float>(), 256, 128, 1, 1, 0);
        ffm_b = use_sh16 ? make_conv_sh16(B, wb, sh, 256, 128, 1, 0) : make_conv(B, wb, sh, 256, 128, 1, 1, 0);
    }
    out_conv = conv_bn(B, "conv_out.conv.conv", "conv_out.conv.bn", 256, 256, 3, 1, 1, use_sh16, wino != 0);
    {   // classifier: 19 rows (f16x3 path: padded to 20 with a zero row, C4 output of 5 groups)
        const auto wc = B.vec("conv_out.conv_out.weight", 19 * 256);
        out_cls = use_sh16 ? make_conv_sh16(B, wc, std::vector<float>(), 19, 256, 1, 0) : make_conv(B, wc, std::vector<float>(), 19, 256, 1, 1, 0);
    }
    auto vecconv = [&](const std::string& conv, const std::string& bn, int o, int i, float*& w, float*& sc, float*& sh) {
        w = B.upload(B.vec(conv + ".weight", (size_t)o * i));
        std::vector<float> s, t;
        bn_fold(B, bn, o, s, t);
        sc = B.upload(s);
        sh = B.upload(t);
    };
    vecconv("cp.conv_avg.conv", "cp.conv_avg.bn", 128, 512, avg_w, avg_scale, avg_shift);
    vecconv("cp.arm16.conv_atten", "cp.arm16.bn_atten", 128, 128, att16_w, att16_scale, att16_shift);
    vecconv("cp.arm32.conv_atten", "cp.arm32.bn_atten", 128, 128, att32_w, att32_scale, att32_shift);
    ffm1_w = B.upload(B.vec("ffm.conv1.weight", 64 * 256));
    ffm2_w = B.upload(B.vec("ffm.conv2.weight", 256 * 64));
    {   // BiSeNet class id -> CelebAMask-HQ id (my_parsing_util.py:19-22,50-54 x global_value_utils.py:49-52)
        const uint8_t lut[19] = {0, 1, 6, 7, 4, 5, 3, 8, 9, 15, 2, 10, 11, 12, 17, 16, 18, 13, 14};
        void* d = B.dalloc(32);
        if (d) (void)hipMemcpy(d, lut, 19, hipMemcpyHostToDevice);
        remap = static_cast<uint8_t*>(d);
    }
    const size_t q = (size_t)mb * (ms / 2) * (ms / 2);   // pixels at 1/2 resolution
    b0 = B.falloc(q * 64);
    b1 = B.falloc(q * 64 / 4);
    b2 = B.falloc(q * 64 / 4);
    f8 = B.falloc(q * 128 / 16);
    f16 = B.falloc(q * 256 / 64);
    f32 = B.falloc(q * 512 / 256);
    vec0 = B.falloc((size_t)mb * 512);
    vec1 = B.falloc((size_t)mb * 512);
    vec2 = B.falloc((size_t)mb * 512);
    splitk_cap = (long long)16 << 20;
    splitk_ws = B.falloc((size_t)splitk_cap);
    amax = static_cast<unsigned*>(B.dalloc(64 * sizeof(unsigned)));
    if (!B.err.empty()) return B.err;
    if (hipDeviceSynchronize() != hipSuccess) return "device sync failed";
    ready = true;
    return "";
}
void BiSeNetModel::destroy() { free_all(allocs); ready = false; }

// The f16x3 trunk: same graph as parse() below, activations in the C4 layout.  A tensor that feeds a conv carries the slot
// its producer records max |value| in; the consuming conv derives the f16 scale from it while it stages the tensor.
std::string BiSeNetModel::parse_sh16(const float* img, uint8_t* labels, float* logits, int Btot, int H, int W, hipStream_t st) {
    struct T { float* p; unsigned* amax; };
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        Ck ck;
        ck(hipMemsetAsync(amax, 0, 64 * sizeof(unsigned), st), "amax reset");
        int ns = 0;
        auto mk = [&](float* ptr) { return T{ptr, amax + ns++}; };
        const int h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4, h8 = H / 8, w8 = W / 8, h16 = H / 16, w16 = W / 16,
                  h32 = H / 32, w32 = W / 32;
        // one conv on the f16x3 kernels over the C4 input (stride 2: conv_sh16.h S2D)
        auto conv = [&](const ConvLayer& L, T in, T out, int hin, int win, int act, const float* res, int in_mode, const char* what) {
            ConvParams p{};
            p.in = in.p;
            p.out = out.p;
            p.B = B;
            p.Cin = L.Cin;
            p.in_mode = in_mode;
            p.H = conv_out_size(L, hin, in_mode);
            p.W = conv_out_size(L, win, in_mode);
            p.Mrows = L.Cout;
            p.bias = L.bias;
            p.act = act;
            p.res = res;
            p.out_amax = out.amax;
            p.partial = splitk_ws;
            p.partial_cap = splitk_cap;
            p.wpk = L.sh_wpk;
            p.wscale = L.sh_wscale;
            p.in_scale_inv = 1.f / SH16_ACT_SCALE;
            p.in_amax = in.amax;
            p.in_c4 = 1;
            if (L.stride == 2) {           // space-to-depth form: output at half the input size
                p.H = hin / 2;
                p.W = win / 2;
                p.s2d_cr = L.s2d_cr;
                p.s2d_phase0 = L.s2d_phase0;
                ck(conv_sh16_s2d_c4(p, L.KS, st), what);
            } else {
                ck(conv_sh16_plain_c4(p, L.KS, st), what);
            }
        };
        ck(stem7x7(img + (size_t)bo * 3 * H * W, stem_w, stem_b, b0, B, H, W, st), "stem");
        T x = mk(b1);
        ck(maxpool3x3s2_c4(b0, x.p, x.amax, B, 64, h2, w2, st), "maxpool");
        float* t0 = b0;
        float* t1 = b0 + (size_t)B * 64 * h4 * w4;
        auto block = [&](const BasicBlockW& bb, T in, float* tmp, float* sc, float* outp, int hin, int win) {
            T t = mk(tmp);
            const int ho = hin / bb.c1.stride, wo = win / bb.c1.stride;
            const float* shortcut = in.p;
            conv(bb.c1, in, t, hin, win, ACT_RELU, nullptr, IN_DIRECT, "bb conv1");
            if (bb.has_down) {               // layers 2-4, first block: conv1 and the 1x1 shortcut are the stride-2 convs
                conv(bb.down, in, T{sc, nullptr}, hin, win, ACT_NONE, nullptr, IN_DIRECT, "bb down");
                shortcut = sc;
            }
            T o = mk(outp);
            conv(bb.c2, t, o, ho, wo, ACT_RELU, shortcut, IN_DIRECT, "bb conv2");
            return o;
        };
        x = block(blk[0], x, t0, nullptr, b2, h4, w4);
        x = block(blk[1], x, t0, nullptr, b1, h4, w4);
        x = block(blk[2], x, t0, t1, b2, h4, w4);
        const T feat8 = block(blk[3], x, t0, nullptr, f8, h8, w8);
        x = block(blk[4], feat8, t0, t1, b2, h8, w8);
        const T feat16 = block(blk[5], x, t0, nullptr, f16, h16, w16);
        x = block(blk[6], feat16, t0, t1, b2, h16, w16);
        const T feat32 = block(blk[7], x, t0, nullptr, f32, h32, w32);
        // ContextPath.forward (model.py:104-125)
        ck(global_avg_pool_c4(feat32.p, vec0, B, 512, h32 * w32, st), "gap32");
        ck(linear(vec0, avg_w, nullptr, avg_scale, avg_shift, vec1, B, 512, 128, 512, 128, ACT_RELU, st), "conv_avg");
        T a = mk(t0);
        conv(arm32_conv, feat32, a, h32, w32, ACT_RELU, nullptr, IN_DIRECT, "arm32 conv");
        ck(global_avg_pool_c4(a.p, vec0, B, 128, h32 * w32, st), "arm32 gap");
        ck(linear(vec0, att32_w, nullptr, att32_scale, att32_shift, vec2, B, 128, 128, 128, 128, ACT_SIGMOID, st), "arm32 atten");
        T s32 = mk(t1);
        ck(chan_affine_c4(a.p, vec2, 0.f, vec1, nullptr, s32.p, s32.amax, B, 128, h32 * w32, st), "feat32_sum");
        T up32 = mk(b1);
        conv(head32, s32, up32, h32, w32, ACT_RELU, nullptr, IN_UP2_NEAREST, "conv_head32");
        a = mk(t0);
        conv(arm16_conv, feat16, a, h16, w16, ACT_RELU, nullptr, IN_DIRECT, "arm16 conv");
        ck(global_avg_pool_c4(a.p, vec0, B, 128, h16 * w16, st), "arm16 gap");
        ck(linear(vec0, att16_w, nullptr, att16_scale, att16_shift, vec2, B, 128, 128, 128, 128, ACT_SIGMOID, st), "arm16 atten");
        T s16 = mk(t1);
        ck(chan_affine_c4(a.p, vec2, 0.f, nullptr, up32.p, s16.p, s16.amax, B, 128, h16 * w16, st), "feat16_sum");
        T cp8 = mk(b2);
        conv(head16, s16, cp8, h16, w16, ACT_RELU, nullptr, IN_UP2_NEAREST, "conv_head16");
        // FeatureFusionModule (model.py:198-210): convblk(cat[feat8, feat_cp8]) as two 1x1 convs
        conv(ffm_a, feat8, T{t0, nullptr}, h8, w8, ACT_NONE, nullptr, IN_DIRECT, "ffm a");
        conv(ffm_b, cp8, T{t1, nullptr}, h8, w8, ACT_RELU, t0, IN_DIRECT, "ffm b");
        ck(global_avg_pool_c4(t1, vec0, B, 256, h8 * w8, st), "ffm gap");
        ck(linear(vec0, ffm1_w, nullptr, nullptr, nullptr, vec1, B, 256, 64, 256, 64, ACT_RELU, st), "ffm conv1");
        ck(linear(vec1, ffm2_w, nullptr, nullptr, nullptr, vec2, B, 64, 256, 64, 256, ACT_SIGMOID, st), "ffm conv2");
        T fo = mk(t0);
        ck(chan_affine_c4(t1, vec2, 1.f, nullptr, nullptr, fo.p, fo.amax, B, 256, h8 * w8, st), "ffm out");
        // BiSeNetOutput (model.py:43-46)
        T oc = mk(t1);
        conv(out_conv, fo, oc, h8, w8, ACT_RELU, nullptr, IN_DIRECT, "conv_out.conv");
        conv(out_cls, oc, T{b1, nullptr}, h8, w8, ACT_NONE, nullptr, IN_DIRECT, "conv_out.conv_out");       // [B,5,h8,w8,4]
        ck(bilinear_argmax(b1, labels + (size_t)bo * H * W, logits ? logits + (size_t)bo * 19 * H * W : nullptr, remap, B, h8,
                           w8, H, W, st, 1), "bilinear argmax");
        if (!ck.err.empty()) return ck.err;
        if (ns > 64) return "BiSeNet: out of amax slots";
    }
    return "";
}

// BiSeNet.forward (model.py:241-254) -> [0] only (my_parsing_util.py:45), argmax + label swap (:46-54)
std::string BiSeNetModel::parse(const float* img, uint8_t* labels, float* logits, int Btot, int H, int W, hipStream_t st) {
    if (!ready) return "BiSeNet not finalized";
    if (H % 32 || W % 32 || H > max_size || W > max_size || H < 64 || W < 64) return "H, W must be multiples of 32 within max_size";
    if (use_sh16) return parse_sh16(img, labels, logits, Btot, H, W, st);
    for (int bo = 0; bo < Btot; bo += max_batch) {
        const int B = std::min(max_batch, Btot - bo);
        Ck ck;
        auto SK = [&](ConvOpts o) { o.partial = splitk_ws; o.partial_cap = splitk_cap; o.no_wino = !wino; return o; };
        const int h2 = H / 2, w2 = W / 2, h4 = H / 4, w4 = W / 4, h8 = H / 8, w8 = W / 8, h16 = H / 16, w16 = W / 16,
                  h32 = H / 32, w32 = W / 32;
        // resnet.py:71-80
        ck(stem7x7(img + (size_t)bo * 3 * H * W, stem_w, stem_b, b0, B, H, W, st), "stem");
        ck(maxpool3x3s2(b0, b1, (long long)B * 64, h2, w2, st), "maxpool");
        // BasicBlock (resnet.py:36-48): relu(bn1(conv1)), bn2(conv2), + shortcut, relu
        auto block = [&](const BasicBlockW& bb, const float* x, float* tmp, float* sc, float* out, int hin, int win) {
            ConvOpts r;
            r.act = ACT_RELU;
            ck(run_conv(bb.c1, x, tmp, B, hin, win, SK(r), st), "bb conv1");
            const int ho = hin / bb.c1.stride, wo = win / bb.c1.stride;
            const float* shortcut = x;
            if (bb.has_down) {
                ck(run_conv(bb.down, x, sc, B, hin, win, SK(ConvOpts()), st), "bb down");
                shortcut = sc;
            }
            ConvOpts o;
            o.act = ACT_RELU;
            o.res = shortcut;
            ck(run_conv(bb.c2, tmp, out, B, ho, wo, SK(o), st), "bb conv2");
        };
        // layer1 @1/4 (64ch): x=b1 ; scratch: b2, b0 (b0 is large)
        float* t0 = b0;                       // scratch big enough for any 1/4-res 64ch tensor and below
        float* t1 = b0 + (size_t)B * 64 * h4 * w4;
        block(blk[0], b1, t0, nullptr, b2, h4, w4);
        block(blk[1], b2, t0, nullptr, b1, h4, w4);
        // layer2 @1/8 (128ch) -> feat8
        block(blk[2], b1, t0, t1, b2, h4, w4);
        block(blk[3], b2, t0, nullptr, f8, h8, w8);
        // layer3 @1/16 (256ch) -> feat16
        block(blk[4], f8, t0, t1, b2, h8, w8);
        block(blk[5], b2, t0, nullptr, f16, h16, w16);
        // layer4 @1/32 (512ch) -> feat32
        block(blk[6], f16, t0, t1, b2, h16, w16);
        block(blk[7], b2, t0, nullptr, f32, h32, w32);
        // ContextPath.forward (model.py:104-125)
        ck(global_avg_pool(f32, vec0, B * 512, h32 * w32, st), "gap32");
        ck(linear(vec0, avg_w, nullptr, avg_scale, avg_shift, vec1, B, 512, 128, 512, 128, ACT_RELU, st), "conv_avg");   // avg [B,128]
        ConvOpts relu;
        relu.act = ACT_RELU;
        // arm32 (model.py:67-83)
        ck(run_conv(arm32_conv, f32, t0, B, h32, w32, SK(relu), st), "arm32 conv");
        ck(global_avg_pool(t0, vec0, B * 128, h32 * w32, st), "arm32 gap");
        ck(linear(vec0, att32_w, nullptr, att32_scale, att32_shift, vec2, B, 128, 128, 128, 128, ACT_SIGMOID, st), "arm32 atten");
        ck(chan_affine(t0, vec2, 0.f, vec1, nullptr, t1, (long long)B * 128, h32 * w32, st), "feat32_sum");   // feat*atten + avg_up
        ConvOpts up_relu = relu;
        up_relu.in_mode = IN_UP2_NEAREST;
        ck(run_conv(head32, t1, b1, B, h32, w32, SK(up_relu), st), "conv_head32");                                // feat32_up @1/16
        // arm16
        ck(run_conv(arm16_conv, f16, t0, B, h16, w16, SK(relu), st), "arm16 conv");
        ck(global_avg_pool(t0, vec0, B * 128, h16 * w16, st), "arm16 gap");
        ck(linear(vec0, att16_w, nullptr, att16_scale, att16_shift, vec2, B, 128, 128, 128, 128, ACT_SIGMOID, st), "arm16 atten");
        ck(chan_affine(t0, vec2, 0.f, nullptr, b1, t1, (long long)B * 128, h16 * w16, st), "feat16_sum");     // feat*atten + feat32_up
        ck(run_conv(head16, t1, b2, B, h16, w16, SK(up_relu), st), "conv_head16");                                // feat_cp8 @1/8
        // FeatureFusionModule (model.py:198-210): convblk(cat[feat8, feat_cp8]) as two 1x1 convs
        ck(run_conv(ffm_a, f8, t0, B, h8, w8, SK(ConvOpts()), st), "ffm a");
        ConvOpts fb = relu;
        fb.res = t0;
        ck(run_conv(ffm_b, b2, t1, B, h8, w8, SK(fb), st), "ffm b");                                              // feat
        ck(global_avg_pool(t1, vec0, B * 256, h8 * w8, st), "ffm gap");
        ck(linear(vec0, ffm1_w, nullptr, nullptr, nullptr, vec1, B, 256, 64, 256, 64, ACT_RELU, st), "ffm conv1");
        ck(linear(vec1, ffm2_w, nullptr, nullptr, nullptr, vec2, B, 64, 256, 64, 256, ACT_SIGMOID, st), "ffm conv2");
        ck(chan_affine(t1, vec2, 1.f, nullptr, nullptr, t0, (long long)B * 256, h8 * w8, st), "ffm out");     // feat*atten + feat
        // BiSeNetOutput (model.py:43-46)
        ck(run_conv(out_conv, t0, t1, B, h8, w8, SK(relu), st), "conv_out.conv");
        ck(run_conv(out_cls, t1, b1, B, h8, w8, SK(ConvOpts()), st), "conv_out.conv_out");                       // [B,19,h8,w8]
        ck(bilinear_argmax(b1, labels + (size_t)bo * H * W, logits ? logits + (size_t)bo * 19 * H * W : nullptr, remap, B, h8,
                           w8, H, W, st), "bilinear argmax");
        if (!ck.err.empty()) return ck.err;
    }
    return "";
}

}  // namespace chk
