// aux_models.h -- the three smaller networks either side of the SEAN generator on the CtrlHair path:
// colour/texture MLPs (A14-A15), shape VAE encoder/decoder (A11-A13), BiSeNet face parser (A16-A17).
#pragma once
#include <string>
#include <vector>

#include "net_common.h"

namespace chk {

// ---- colour / texture branch (color_texture_branch/model_eigengan.py, model.py:86-130, predictor_model.py) ----
struct ColorModel {
    bool ready = false;
    int max_batch = 0;
    std::vector<void*> allocs;
    float *g_in_w = nullptr, *g_in_b = nullptr;
    float *g_mid_w[4] = {}, *g_mid_b[4] = {};
    float *sub_U[4] = {}, *sub_L[4] = {}, *sub_mu[4] = {};
    float *d_w[5] = {}, *d_b[5] = {};
    float *p_w[4] = {}, *p_b[4] = {}, *p_scale[3] = {}, *p_shift[3] = {};
    float *wa = nullptr, *wb = nullptr;   // [max_batch][512] scratch
    std::string build(const TensorStore& ts, int max_batch);
    std::string generate(const float* noise, const float* cond, float* code, int B, hipStream_t st);
    std::string encode(const float* code, float* out11, int B, hipStream_t st);
    std::string predict(const float* code, float* out4, int B, hipStream_t st);
    void destroy();
};

// ---- shape branch (shape_branch/model.py, my_torchlib/module.py) ----------------------------------------------
struct LnW { float *gamma = nullptr, *beta = nullptr; };
struct ShapeModel {
    bool ready = false;
    int max_batch = 0;
    static constexpr int S = 256, HAIR_DIM = 16, FACE_DIM = 1024;
    std::vector<void*> allocs;
    ConvLayer enc[2][7];      // [0]=hair, [1]=face
    LnW enc_ln[2][7];
    float *enc_fc_w[2] = {}, *enc_fc_b[2] = {};
    float *dec_in_w[2] = {}, *dec_in_b[2] = {};
    ConvLayer dec[2][7], dec_out[2];
    LnW dec_ln[2][7];
    // decoder layers 1..6 on the f16x3 split-operand kernels (conv_sh16.h): packed weights, per-row inverse scales, and the
    // power-of-two SH16 scale of each LayerNorm output (from the bound sqrt(C*HW) * max|gamma| + max|beta|: cannot saturate)
    bool use_sh16 = true;
    int wino = 1;                   // option "aux.wino": exact-f32 kernels, 3x3 stride-1 convs as Winograd F(2x2,3x3) where the tiles fit (net_common.h run_conv)
    float *dec_sh[2][7] = {}, *dec_ws[2][7] = {};
    float dec_ln_scale[2][7] = {};
    // encoder layers (k4 s2 convs, 128^2 .. 2^2 outputs) in the space-to-depth form of the f16x3 kernels (conv_sh16.h S2D):
    // SH16 inputs (scale 2^14: one-hot and sin / cos channels), LayerNorm outputs SH16 with their static scales
    ConvLayer dec_out_sh[2];        // the output convs (32 -> 1 / 18, rows padded to 4 / 20) for the f16x3 kernels: C4 logits
    static constexpr int ENC_S2D = 7;          // layers 0 .. ENC_S2D-1 on the S2D kernels (the rest on the exact-f32 kernels)
    ConvLayer enc_s2d[2][7];
    float enc_ln_scale[2][7] = {};
    static constexpr float ENC_IN_SCALE = 16384.f;
    float* pos = nullptr;      // [40][S*S]
    // exact-f32 path: layer 0 of both encoders as a label table (misc_kernels.hip shape_enc_l0): the positional channels' part of the
    // conv + bias per output element, and the mask channels' weights as [encoder][tap][label row][channel]
    float* enc0_pc[2] = {nullptr, nullptr};
    float* enc0_tab = nullptr;
    std::vector<float> enc0_tab_host;       // (build() only)
    int enc_l0_lut = 1;        // option "shape.enc_lut": 0 = layer 0 through the conv kernel on materialised one-hot + positional inputs
    float *in_hair = nullptr, *in_face = nullptr, *bufa = nullptr, *bufb = nullptr, *bufc = nullptr, *lnpart = nullptr,
          *codecat = nullptr, *splitk_ws = nullptr;
    long long splitk_cap = 0;
    // the two encoders (decoders) are independent chains of small launches: the hair one runs on a side stream with its own workspace
    int overlap = 1;           // option "shape.overlap"
    float *bufa2 = nullptr, *bufb2 = nullptr, *bufc2 = nullptr, *lnpart2 = nullptr, *splitk_ws2 = nullptr;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string build(const TensorStore& ts, int max_batch);
    std::string encode(const uint8_t* labels, float* hair_code, float* face_code, int B, hipStream_t st);
    // decode: any of hair_logit/face_logit/labels/probs may be null; hair_code may be null when only the face is wanted
    std::string decode(const float* hair_code, const float* face_code, float* hair_logit, float* face_logit,
                       uint8_t* labels, float* probs, int B, hipStream_t st);
    std::string combine(const float* hair_logit, const float* face_logit, uint8_t* labels, float* probs, int B,
                        hipStream_t st);
    void destroy();
  private:
    std::string run_encoder(int which, const float* in, float* code, int B, hipStream_t st, int set);
    // logit: NCHW [B][1 | 18][S*S]; f16x3 path: C4 [B][1 | 5][S*S][4] (rows padded)
    std::string run_decoder(int which, const float* code, int code_dim, float* logit, int B, hipStream_t st, int set);
};

// ---- BiSeNet (external_code/face_parsing/model.py, resnet.py, my_parsing_util.py) ------------------------------
struct BasicBlockW { ConvLayer c1, c2, down; bool has_down = false; };
struct BiSeNetModel {
    bool ready = false;
    int max_batch = 0, max_size = 0;
    std::vector<void*> allocs;
    float *stem_w = nullptr, *stem_b = nullptr;
    BasicBlockW blk[8];
    ConvLayer arm16_conv, arm32_conv, head32, head16, ffm_a, ffm_b, out_conv, out_cls;
    // 1x1 convs on pooled vectors as GEMV (+ folded BN)
    float *avg_w = nullptr, *avg_scale = nullptr, *avg_shift = nullptr;
    float *att16_w = nullptr, *att16_scale = nullptr, *att16_shift = nullptr;
    float *att32_w = nullptr, *att32_scale = nullptr, *att32_shift = nullptr;
    float *ffm1_w = nullptr, *ffm2_w = nullptr;
    uint8_t* remap = nullptr;
    // f16x3 trunk (option "bisenet.f16x3", default on): every stride-1 conv on the split-operand f16 MFMA kernels with the
    // f32 activations kept in the C4 layout and split while they are staged (conv_sh16.h, INC4); the stride-2 convs stay on
    // the exact-f32 kernels (C4 in / out).  `amax`: one device slot per activation tensor (sh16.h).
    bool use_sh16 = true;
    int wino = 1;                   // option "aux.wino" (see ShapeModel)
    unsigned* amax = nullptr;
    float *b0 = nullptr, *b1 = nullptr, *b2 = nullptr, *f8 = nullptr, *f16 = nullptr, *f32 = nullptr, *vec0 = nullptr,
          *vec1 = nullptr, *vec2 = nullptr, *splitk_ws = nullptr;
    long long splitk_cap = 0;
    std::string build(const TensorStore& ts, int max_batch, int max_size);
    std::string parse(const float* img, uint8_t* labels, float* logits, int B, int H, int W, hipStream_t st);
    void destroy();
  private:
    std::string parse_sh16(const float* img, uint8_t* labels, float* logits, int B, int H, int W, hipStream_t st);
};

}  // namespace chk
