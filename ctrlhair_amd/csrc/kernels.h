// kernels.h -- launchers of the non-MFMA kernels (sean_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace chk {
hipError_t label_downsample(const uint8_t* in, uint8_t* out, int B, int S, int r, hipStream_t s);
// need (optional, [B][H][W]): pixels with need == 0 are not written (ace_sparse.h: nothing reads them)
hipError_t onehot_conv3x3(const uint8_t* lab, const float* table, const float* bias, float* out, int B, int H, int W,
                          int K, int relu, hipStream_t s, int c4 = 0, const uint8_t* need = nullptr, int kout = 0);
// kout > K (NCHW output only): the output tensor has kout channel planes per sample; the table channels fill the first K.
// label_onehot_planes writes planes k0 .. k0 + 19 of such a tensor: plane k0 + j = (label == j), plane k0 + 19 = 0 (the Winograd ACE
// kernel's style k-steps, conv_wino.h)
hipError_t label_onehot_planes(const uint8_t* lab, float* out, int B, int H, int W, int kout, int k0, hipStream_t s);
// Winograd ACE path: the SPADE hidden activations (K = 128 table channels, relu) + optionally the 20 one-hot planes behind them, written
// only at the pixels some boundary quad's 4 x 4 patch reads (u5: the interior map of ace_classify, 255 = boundary pixel; nullptr:
// every quad is a boundary quad).  out has kout planes per sample, each H rows of `pitch` floats with image column x at x + xoff
// (pitch = 0: plain [H][W] planes; xoff > 0: the padded layout of the Winograd ACE kernels, conv_wino.h WINO_AXOFF -- the columns
// left and right of the image must hold zeros, the kernel never writes them: the caller clears them).  Shapes: spade_hidden_wq_supported (H, W multiples of 32).
bool spade_hidden_wq_supported(int H, int W);
// Second half of an ACE whose SPADE gamma / beta conv ran as a PLAIN conv (tiny levels at small batches: sean_model.cpp ace()):
//   gb [B][rowsP][H*W] holds the raw conv sums in the row order of the direct kernels' packed image (wave tile of 64 rows = gamma rows |
//   beta rows of 32 channels); adds the blended biases and the style-LUT gathers (lut [B*19][9][2][C] or null), modulates
//   act((bn_a x + nv noise + bn_d)(1 + gamma) + beta) -- normalization.py:111-112,117-153,172-187; the arithmetic of ace_epilogue_f32.
hipError_t ace_finish_f32(const float* gb, int rowsP, const float* x, int x_up, const float* bias_g, const float* bias_b, const float* bn_a,
                          const float* bn_d, const float* nv, const float* noise, long long noise_bstride, const uint8_t* lab, const float* lut,
                          float* out, int B, int C, int H, int W, int act, hipStream_t s);
hipError_t spade_hidden_wq(const uint8_t* lab, const uint8_t* u5, const float* table, const float* bias, float* out, int B, int H, int W,
                           int kout, int onehot, hipStream_t s, int pitch = 0, int xoff = 0, const int* skip_if_patch = nullptr);
// the same activations as pre-gathered 4 x 4 patches of the boundary quads, chunk by chunk of 64 (conv_wino.h WinoAceParams::patch); both
// kernels read the device flag *mode (wino_chunk_base): exactly one of them does the work
hipError_t spade_hidden_patch(const uint8_t* lab, const unsigned* gq, const int* gq_n, int gq_cap, const int* chunk_base, const int* mode,
                              const float* table, const float* bias, float* patch, int B, int H, int W, int kout, hipStream_t s);
hipError_t wino_chunk_base(const int* gq_n, int B, int cap_chunks, int* chunk_base, int* mode, hipStream_t s);
// `scale`: power-of-two scale of an SH16 output / input tensor (sh16.h)
hipError_t onehot_conv3x3_sh16(const uint8_t* lab, const float* table, const float* bias, void* out, int B, int H, int W,
                               int K, int relu, float scale, hipStream_t s, int bf16 = 0, const uint8_t* need = nullptr,
                               const int* tile_cnt = nullptr,    // tile_cnt: boundary pixels per tile of 32 x 16 (tile-skip mode)
                               int need_impl = 0);               // need-masked launches: 0 = compacting kernel at W >= 512,
                                                                 // 1 = never, 2 = always (A/B, tools/onehot_bench.hip)
// amax: device slot of a dynamically scaled tensor (sh16.h), null = static scale
hipError_t sh16_decode(const void* in, float* out, int B, int C, long long HW, float scale, const unsigned* amax,
                       hipStream_t s, int bf16 = 0);
hipError_t fc_mu(const float* codes, const float* Wt, const float* bias, float* mu_img, int B, int Npad, hipStream_t s,
                 float* mu_rows = nullptr, int sh16 = 0, int bs = 19, float scale = 1.f, unsigned* amax = nullptr,
                 int pass = 0, int bf16 = 0);
hipError_t fc_mu_batched(const float* codes, const float* const* Wts, const float* const* biases, float* mu_base,
                         long long mu_stride, int n_aces, int B, int Npad, int bs, float scale, unsigned* amax_slots, int pass,
                         int bf16, hipStream_t s, int sh16 = 1);
// w4 (C4 path): the same weights packed [Cin/4][tap][co][4 channels] for scalar loads
hipError_t conv_img_tanh(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int W,
                         hipStream_t s, int c4 = 0, const float* w4 = nullptr);
hipError_t c4_decode(const float* in, float* out, int B, int C, long long HW, hipStream_t s);
hipError_t gen_noise(float* out, long long n, uint64_t seed, hipStream_t s);
// mfma_peak.hip: sustained MFMA-only rate of the device (kind 0: f32 32x32x2, 1: f16 32x32x16), ~ms_target ms, synchronises
hipError_t mfma_peak(int kind, int ms_target, double* tflops, hipStream_t s);
// misc_kernels.hip
// instance-norm outputs are bounded by sqrt(HW): their SH16 scale is instnorm_sh16_scale(HW), no saturation possible
float instnorm_sh16_scale(int HW);
// scratch (optional, >= 64 KiB of floats): enables the sliced two-kernel form for tensors with few (sample, 8-channel group)
// pairs and large planes, where one block per pair would leave most CUs idle
hipError_t instnorm_act(float* x, int planes, int HW, float eps, int act, hipStream_t s, void* sh16 = nullptr, int C = 0,
                        float* scratch = nullptr);
// ConvTranspose2d(k3, s2, p1, op1) as four phase GEMMs (misc_kernels.hip): the four shifted views of x [B][C][H][W] as xs [B][4][C][H][W]
// (shift order (0,1) | (0,0) | (1,0) | (1,1)); InstanceNorm + activation over the phase planes t [4][B][C][H][W] (+ bias), written
// depth-to-space into out [B][C][2H][2W]
hipError_t convt_shift4(const float* x, float* xs, int B, int C, int H, int W, hipStream_t s);
hipError_t instnorm_act_d2s(const float* t, const float* bias, float* out, int B, int C, int H, int W, float eps, int act, hipStream_t s);
hipError_t instnorm_c4_to_sh16(const float* x_c4, int B, int C, int HW, float eps, int act, void* sh16, hipStream_t s,
                               float* scratch = nullptr);
hipError_t layernorm_act(float* x, const float* gamma, const float* beta, float* part, int B, int C, int HW, float eps,
                         int act, hipStream_t s);
// LayerNorm + activation with layout conversion: in NCHW / C4 -> out SH16 (scaled) / NCHW f32 (shape decoder, f16x3 convs)
hipError_t layernorm_act_conv(const float* x, int in_c4, void* out, int out_sh16, float out_scale, const float* gamma,
                              const float* beta, float* part, int B, int C, int HW, float eps, int act, hipStream_t s);
hipError_t region_mean(const float* codes, const uint8_t* lab, float* out, int B, int F, int h, int w, int S,
                       hipStream_t s, int c4 = 0);
hipError_t maxpool3x3s2(const float* in, float* out, long long planes, int H, int W, hipStream_t s);
hipError_t global_avg_pool(const float* in, float* out, int planes, int HW, hipStream_t s);
hipError_t chan_affine(const float* in, const float* sc, float sc_add, const float* sh, const float* other, float* out,
                       long long planes, int HW, hipStream_t s);
hipError_t stem7x7(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s);
hipError_t conv3x3_c3_reflect(const float* in, const float* w, const float* bias, float* out, int B, int Cout, int H, int W,
                              hipStream_t s);
hipError_t bilinear_argmax(const float* lg, uint8_t* out, float* logits_out, const uint8_t* remap, int B, int h, int w,
                           int H, int W, hipStream_t s, int c4 = 0);
// C4-layout variants (f16x3 BiSeNet trunk); `amax`: device slot receiving max |out| * SH16_ACT_SCALE, or null
hipError_t maxpool3x3s2_c4(const float* in_nchw, float* out_c4, unsigned* amax, int B, int C, int H, int W, hipStream_t s);
hipError_t global_avg_pool_c4(const float* in, float* out, int B, int C, int HW, hipStream_t s);
hipError_t chan_affine_c4(const float* in, const float* sc, float sc_add, const float* sh, const float* other, float* out,
                          unsigned* amax, int B, int C, int HW, hipStream_t s);
hipError_t shape_softmax(const float* hair, const float* face, uint8_t* lab, float* probs, int B, int HW, hipStream_t s,
                         int c4 = 0);
hipError_t c4_rows_to_nchw(const float* in, float* out, int B, int C, int Cpad, int HW, hipStream_t s);
// layer 0 of both shape encoders from the label map: out = posconst + 16 table rows per pixel (misc_kernels.hip); either output may be null
hipError_t shape_enc_l0(const uint8_t* lab, const float* tab, const float* pc_hair, const float* pc_face, float* out_hair, float* out_face,
                        int B, int S, hipStream_t s);
hipError_t shape_inputs(const uint8_t* lab, const float* pos, float* hair_in, float* face_in, int B, int HW,
                        hipStream_t s);
hipError_t shape_inputs_sh16(const uint8_t* lab, const float* pos, void* hair_in, void* face_in, int B, int HW, float scale,
                             hipStream_t s);      // SH16, 48 / 64 channels
hipError_t linear(const float* x, const float* W, const float* bias, const float* scale, const float* shift, float* out,
                  int B, int K, int O, int ldx, int ldo, int act, hipStream_t s);
// out[n][rows] = W[rows][512] x[n][:]: the style LUT of one ACE for N <= 64 (sample, label) columns on the f32 matrix cores (misc_kernels.hip)
hipError_t lut_gemv_mfma(const float* x, const float* W, float* out, int N, int rows, hipStream_t s);
hipError_t subspace_add(float* h, const float* z, int zld, const float* U, const float* L, const float* mu, int B, int D,
                        int Z, hipStream_t s);
// poisson_kernels.hip: blending step after the generator (hair_editor.py:285-310, poisson_blending.py:29-87)
size_t poisson_workspace_bytes(int H, int W);
hipError_t poisson_blend(const uint8_t* src_hwc, const uint8_t* tgt_hwc, const uint8_t* mask, uint8_t* out_hwc, int H, int W,
                         int with_gamma, int max_iters, double rel_tol, void* ws, int* iters_out, hipStream_t s);
hipError_t blend_mask(const uint8_t* target_parsing, const uint8_t* face_parsing, uint8_t* out, int H, int W, hipStream_t s);

}  // namespace chk
