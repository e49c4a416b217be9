/*
 * ctrlhair_hip.h -- C ABI of libctrlhair_hip.so: the MI355X (gfx950) implementation of the CtrlHair
 * convolutional GAN *inference* forward path.
 *
 * The reference (XuyangGuo/CtrlHair) is pure Python on torch.nn; it has no FFI of its own.  Each entry
 * point below therefore replaces a Python call site of the reference (cited per function), and is what a
 * ctypes binding added to the reference would bind.  Plain pointers and sizes only;
 * no torch types.  All device pointers are HIP device memory owned by the caller; all work is enqueued on
 * the caller's stream; no entry point synchronises the device except ch_finalize()/ch_destroy().
 *
 * Error convention: int status, 0 = ok, non-zero = failure; message via ch_last_error().  No C++ exception
 * crosses the ABI.  A handle is bound to one device and is not thread-safe; different handles are independent.
 */
#ifndef CTRLHAIR_HIP_H
#define CTRLHAIR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CH_ABI_VERSION 1

typedef struct ch_handle ch_handle;
typedef void* ch_stream_t;              /* hipStream_t */

enum ch_status { CH_OK = 0, CH_ERR_ARG = 1, CH_ERR_HIP = 2, CH_ERR_STATE = 3, CH_ERR_WEIGHTS = 4 };

/* which network a tensor belongs to (one handle can hold all of them) */
enum ch_model {
    CH_MODEL_SEAN = 0,         /* sean_codes/models/networks/generator.py: SPADEGenerator (+Zencoder) */
    CH_MODEL_SHAPE = 1,        /* shape_branch/model.py: Generator (hair/face MaskEncoder + MaskDecoder), cfg 054 */
    CH_MODEL_COLOR = 2,        /* color_texture_branch: EigenGenerator "gen.*", Discriminator "dis.*", rgb Predictor "rgb.*" */
    CH_MODEL_BISENET = 3       /* external_code/face_parsing/model.py: BiSeNet(19) */
};

enum ch_dtype { CH_F32 = 0, CH_I64 = 1 };

int  ch_abi_version(void);

/* Replaces model construction + .cuda(): sean_codes/models/networks/__init__.py:39-51 (create_network),
 * hair_editor.py:45-51.  Binds the handle to HIP device `device`. */
int  ch_create(int device, ch_handle** out);
void ch_destroy(ch_handle* h);
const char* ch_last_error(const ch_handle* h);

/* Replaces util/util.py:202-208 (load_network -> net.load_state_dict): hand over one state-dict entry under its
 * reference key name (e.g. "up_0.conv_0.weight_orig", "head_0.ace_0.fc_mu3.weight").  `host` is host memory,
 * copied before return.  Unknown names are kept and ignored at finalize (the reference state dict carries unused
 * buffers: Spade.param_free_norm.*, num_batches_tracked). */
int  ch_load_tensor(ch_handle* h, int model, const char* name, const void* host, int dtype,
                    const int64_t* shape, int ndim);

/* Options, set before ch_finalize.  "sean.f16x3" (default 0) selects the arithmetic of the SEAN generator's MFMA convs:
 *   0  f32 throughout on the f32 matrix cores (v_mfma_f32_16x16x4_f32 / 32x32x2_f32: every product and every sum an IEEE f32
 *      operation); with "sean.wino" >= 1 the 3x3 convs are evaluated as Winograd convolutions (ctrlhair_amd/csrc/conv_wino.h,
 *      conv_wino4.h: f32 operands and f32 accumulation, but TRANSFORMED operands -- U = G g G^T is computed in double and rounded
 *      once to f32, the input / output transforms are f32 adds / fmas -- so this is not a re-association of the direct sum and
 *      carries its own rounding error: F(2x2,3x3) <= ~4e-6 against the direct evaluation on the generator output, F(4x4,3x3)
 *      (the default for the ResBlock convs and the SPADE convs up to 64 pixels) <= ~3e-5; the contract tolerance is 1e-3),
 *      with 0 directly (conv_mfma.h: the reference conv2d's products and sums);
 *   1  f16 matrix cores with the 3-term split-operand scheme of ctrlhair_amd/csrc/conv_sh16.h: f32-class accuracy, f32
 *      accumulation, activations between ACE and conv stored as f16 hi/lo pairs;
 *   2  f16 matrix cores, single term: operands rounded to f16, f32 accumulation and f32 normalisation / modulation
 *      (the reduced-precision configuration of BASELINE.json configs[4]; tolerance 5e-2);
 *   3  bf16 matrix cores (v_mfma_f32_32x32x16_bf16), single term: operands rounded to bf16, f32 accumulation and f32
 *      normalisation / modulation -- configs[4] as written ("bf16 MFMA conv path"; tolerance 5e-2).
 * "sean.ahead" (default 8): run-ahead mode -- when a batch chunk holds at most `value` x 512x512 pixels, the kernels of the
 *   18 ACE layers that depend only on the label map and the style codes (label tables, fc_mu, style LUTs) run on an internal
 *   side stream into per-layer buffers, joined to `stream` by events (interactive latency: 3.3 -> 2.65 ms at 256x256; still
 *   1.6 % at 8 x 512x512).  Handles sized for larger chunks run only the style LUT builds (small GEMMs) ahead and keep the
 *   HBM-write-bound label-table kernels inline (1 % at 16 x 512x512; with the label tables ahead too: no gain).  0 = off.
 * "shape.f16x3" (default 1): the shape VAE's convs run on the same f16x3 split-operand kernels -- all encoder layers (k4,
 *   stride 2: space-to-depth staging), decoder layers 1-6 and the output convs; LayerNorm outputs and the one-hot / sin-cos
 *   inputs are bounded, so their scales are static; 0 = every conv on the exact-f32 kernels.
 * "bisenet.f16x3" (default 1): BiSeNet's convs on the f16x3 kernels; its f32 activations stay in the C4 layout and are split
 *   into f16 pairs while staged, with the scale derived from the maximum the producing kernel recorded; 0 = exact-f32 kernels.
 *   (The Zencoder follows "sean.f16x3".)
 * "aux.wino" (default 1): the exact-f32 kernels of the shape VAE and BiSeNet ("shape.f16x3" / "bisenet.f16x3" = 0, and the layers those
 *   options leave on them) run their 3x3 stride-1 convs as Winograd F(2x2,3x3) wherever the output fits the kernel's tiles; 0 = direct.
 * "sean.wino" (default 2; "sean.f16x3" = 0 only): 1 = the ResBlock 3x3 convs, the SPADE gamma/beta convs and the style convs as
 *   Winograd F(2x2,3x3) on the f32 matrix cores, the learned 1x1 shortcuts on the pointwise kernel of conv_pw.h; 2 = in addition the
 *   ResBlock convs from 32 pixels up as Winograd F(4x4,3x3) (conv_wino4.h: 36 instead of 64 products per 4 x 4 pixels); 0 = direct.
 * "sean.wino4_ace" (default 64; with "sean.wino" = 2): the SPADE / style convs of the levels up to this many pixels (multiples of 32)
 *   run as dense F(4x4,3x3) over every tile instead of F(2x2,3x3) over the boundary quads; 0 = never.
 *   Both F(4x4,3x3) choices are made per call: with fewer tasks of 32 x 32 pixels than a round of F(2x2,3x3) tasks would need CUs (single
 *   images, small batches of small images) the F(2x2,3x3) kernels run instead (wino4_pays, conv_wino4.h); "sean.wino4_force" = 1 (any
 *   time) switches that rule off.
 * "sean.edge" (default 1; before ch_finalize; exact-f32 Winograd path and the f16x3 / f16 / bf16 paths with pixel-level compaction): a boundary pixel whose 5 x 5 label neighbourhood is five uniform
 *   columns (or rows) A..A B..B -- one straight, axis-aligned piece of a region border -- takes gamma / beta from the row of its code
 *   (orientation, A, B, number of A columns: 2888 rows per ACE, built at ch_finalize in double) plus three column / row sums of the style
 *   LUT, in the interior pass, on the levels of 128 pixels and more; only corners, curved pieces and the image frame go through the
 *   boundary conv (csrc/ace_sparse.h).  Same real number, another association of the f32 sums (<= 1e-6 against "sean.edge" = 0;
 *   f16x3: <= 5e-6; on the single-term f16 / bf16 paths the table rows are exact f32 where the conv they replace is not).
 *   Costs 52 MB of tables at ngf = 64.  0 = every non-interior pixel through the conv.
 * "sean.convt_gemm" (default 1; before ch_finalize; exact-f32 path): the Zencoder's ConvTranspose2d(128, 256, k3, s2, p1, op1)
 *   (architecture.py:167-170) as four phase GEMMs over shifted views of its input -- 9 products per 2 x 2 outputs and channel pair -- with the
 *   InstanceNorm + lrelu that follows reading the phase planes; calls with fewer than 16384 input pixels and 0 = four Winograd F(2x2,3x3)
 *   phase convs (16 products).  Same sums in another order (<= 1e-6 on the style codes).
 * "sean.patch" (default 1; before ch_finalize): when a level is left with few boundary quads (at most 32 chunks of 64 per sample) their
 *   hidden-activation patches are written pre-gathered, in the conv kernel's stage layout, instead of being fetched piecewise from the
 *   planes (csrc/conv_wino.h WinoAceParams::patch); decided per level and call on the device; bit-identical.  0 = planes only.
 * "sean.batch_invariant" (default 0; any time; exact-f32 path): by default several choices follow the number of tasks of a call, i.e. its
 *   batch size -- F(4x4,3x3) vs F(2x2,3x3) (wino4_pays), split-K of launches with few tasks, sample-pair tiles of the 16-pixel level, the
 *   GEMV / tiny-level routes of interactive batches -- so the same sample rendered alone and inside a batch differs by the rounding of two
 *   associations of the same f32 sums (measured <= 3e-5 at 512 x 512).  1 = every such choice is made as for a large batch: sample i
 *   alone == sample i in any batch of the same handle, bit for bit (tests/test_hip_sean_generator.py); costs latency on small calls
 *   (bench.py reports both).
 * "sean.wino4v" (default 1; before ch_finalize; with "sean.wino" = 2): F(4x4,3x3) layers with at least 512 GEMM rows at up to 64 x 64
 *   pixels -- the ResBlock convs of G_middle / up_0 and every SPADE / style conv that "sean.wino4_ace" selects -- and the Zencoder's
 *   256 -> 512 conv read their input pre-transformed (V = B^T d B, written once by an extra bandwidth-bound pass, csrc/conv_wino4v.h)
 *   instead of transforming it again in every row tile; same arithmetic in the same order: bit-identical results.  Costs a workspace
 *   of 9 bytes per input element of the largest such layer (604 MB at max_batch 16, 512 x 512).  0 = transform inside the conv kernel.
 * "sean.overlap" (default 0; before ch_finalize): number of CUs given to CU-masked side streams on which the interior passes and
 *   label-table kernels run beside the convs (with dynamic task claiming in the Winograd kernels).  Bit-identical results; measured
 *   SLOWER than the serial order at every setting on MI355X: kept as an option, not used.
 * "sean.lut_grouped" (default 1; exact-f32 path, calls with more than 64 (sample, label) columns): the style LUTs of all styled ACEs
 *   of a chunk come from ONE grouped GEMM launch at its start (csrc/conv_pw.h); 0 = one launch of the generic 1x1 kernel per ACE.
 * "sean.hidden_wq" (default 1; any time): every Winograd ACE level (32 pixels and more) takes the SPADE hidden activations and the
 *   one-hot planes from one persistent kernel (csrc/sean_kernels.hip spade_hidden_wq).  1 = on the gather levels it writes only the
 *   64-byte pixel groups that a boundary quad's patch touches; 0 = the same kernel over every pixel (bit-identical results).  The dense
 *   F(4x4,3x3) levels ("sean.wino4_ace") read every pixel either way.  (The two-kernel route of round 4 -- label table + one-hot kernel --
 *   serves the levels below 32 pixels only.)
 * "sean.wino_gather" (default 1): the Winograd ACE kernel takes tasks of 64 consecutive boundary quads of a sample and fetches each
 *   quad's own 4 x 4 patch (csrc/conv_wino.h); 0 = tasks per tile of 32 x 16 / 32 x 32 pixels (bit-identical results).
 * "sean.wino_th": tile height 16 / 32 of the tile mode (0 = chosen per resolution level).
 * "sean.sparse" (default 1): the exact SPADE-interior reduction (csrc/ace_sparse.h).  May be switched off (and back on) after
 *   ch_finalize; a handle finalised with 0 has no classification buffers and rejects 1 afterwards (CH_ERR_STATE).
 * "sean.sparse_min", "sean.sparse_th", "sean.sh16_compact": tuning knobs of that reduction (before ch_finalize).
 * "sean.dbg": profiling switches.  The bits that skip work (wrong results) or select superseded kernel versions exist only in
 *   libraries built with -DCH_ABLATE (make -C ctrlhair_amd/csrc ABLATE=1); the default build rejects them (CH_ERR_ARG). */
int  ch_set_option(ch_handle* h, const char* key, int value);

/* Fold + pack + upload the loaded tensors: spectral-norm sigma (torch spectral_norm eval semantics,
 * architecture.py:42-46), eval-BN running stats -> per-channel affine (sync_batchnorm/batchnorm.py:52-55),
 * sigmoid(blending) folded into the SPADE / style weights (normalization.py:177-181), one-hot convs -> label
 * LUTs, MFMA operand layouts.  Sizes the workspace arena for images up to max_size x max_size in chunks of
 * max_batch.  ngf is inferred from the tensors.  For CH_MODEL_SHAPE / CH_MODEL_COLOR max_size is ignored.
 * Synchronises the device. */
int  ch_finalize(ch_handle* h, int model, int max_batch, int max_size);

/* Floats of noise per sample at image side S: sum over the 18 ACE layers (execution order: ace_s, ace_0, ace_1
 * per block) of (S/res_div)^2 -- the planes normalization.py:111 draws as randn(B, W, H, 1). */
size_t ch_sean_noise_floats(const ch_handle* h, int S);

/* Replaces Pix2PixModel.forward(data, mode='UI_mode') -> SPADEGenerator.forward
 * (sean_codes/models/pix2pix_model.py:59-68,119-144; generator.py:72-109; architecture.py:69-96;
 * normalization.py:108-189), batched: every sample b gets the UI_mode treatment the reference gives sample 0.
 *   labels : device uint8  [B,S,S]       CelebAMask-HQ ids 0..18 (the one-hot of pix2pix_model.py:133-138 is
 *                                        never materialised)
 *   codes  : device float  [B,19,512]    per-region style codes (obj_dic[str(j)]['ACE'])
 *   noise  : device float  [B,noise_floats(S)] explicit noise planes n_k[b][w][h], or NULL to draw them on
 *            device from `seed` (counter-based generator; the reference draws torch.randn)
 *   out    : device float  [B,3,S,S]     image in [-1,1] (tanh)
 * S must be a multiple of 32 with S <= max_size.  Asynchronous on `stream`. */
int  ch_sean_generate(ch_handle* h, const uint8_t* labels, const float* codes, const float* noise, uint64_t seed,
                      float* out, int B, int S, ch_stream_t stream);

/* The noise planes ch_sean_generate(noise = NULL, seed) draws on device, written out: noise [B, noise_floats(S)] (device).
 * ch_sean_generate(..., noise = these planes, ...) reproduces the NULL call bit for bit; the planes are i.i.d. N(0,1) from a
 * counter-based generator (the reference draws torch.randn from an unseeded global generator, normalization.py:111). */
int  ch_sean_draw_noise(ch_handle* h, uint64_t seed, float* noise, int B, int S, ch_stream_t stream);

/* Replaces Pix2PixModel.forward(data, mode='style_code') -> Zencoder.forward
 * (pix2pix_model.py:69-72; architecture.py:177-207; callers hair_editor.py:149-157 get_code, :208-231):
 * conv stack (reflection-padded 3x3, two stride-2 convs, ConvTranspose2d, InstanceNorm + LeakyReLU, tanh) at
 * S/2 resolution, then per-region average pooling with the label map nearest-down-sampled to S/2.
 *   img    : device float [B,3,S,S] in [-1,1]        labels : device uint8 [B,S,S]
 *   codes  : device float [B,19,512] (rows of absent regions are 0)
 * Requires the Zencoder.* tensors to have been loaded before ch_finalize. */
int  ch_sean_encode(ch_handle* h, const float* img, const uint8_t* labels, float* codes, int B, int S,
                    ch_stream_t stream);
/* The same encoder in two calls, for callers that compute the label map while the convolutions run (the labels only enter
 * the region means, architecture.py:185-205): ch_sean_encode_features runs the convolutional part and keeps the feature map
 * in the handle's workspace; ch_sean_encode_regions reduces it to codes.  One chunk (B <= max_batch); the second call must
 * follow the first with the same B, S (else CH_ERR_HIP), with no other SEAN call on the handle in between, and be ordered
 * after it (same stream, or an event).  ch_sean_encode(img, labels, codes) == features(img); regions(labels, codes). */
int  ch_sean_encode_features(ch_handle* h, const float* img, int B, int S, ch_stream_t stream);
int  ch_sean_encode_regions(ch_handle* h, const uint8_t* labels, float* codes, int B, int S, ch_stream_t stream);

/* ---- colour / texture branch (three MLPs on 512-d hair style codes) -------------------------------------------
 * Tensor names: the reference state-dict keys prefixed "gen." (EigenGenerator, model_eigengan.py:34-84), "dis."
 * (Discriminator used as encoder, model.py:86-130) and "rgb." (Predictor p004, predictor_model.py:14-41).
 * ch_color_generate replaces feature_generator(data)['code'] (ui/backend.py:166-169, solver.py:78-83):
 *   noise [B,8], cond [B,5] = cat(noise_curliness[1], rgb_mean[3], pca_std[1]) (model_eigengan.py:66-74) -> code [B,512]
 * ch_color_encode replaces feature_encoder({'code'}) (ui/backend.py:103-105): raw net output [B,11]; columns
 *   0 = adv, 1..8 = noise, 9 = noise_curliness, 10 unused (model.py:112-127 slices them on the host).
 * ch_color_predict replaces feature_rgb_predictor({'code'}) (ui/backend.py:96): [B,4] = rgb_mean[3], pca_std[1]. */
int  ch_color_generate(ch_handle* h, const float* noise, const float* cond, float* code, int B, ch_stream_t stream);
int  ch_color_encode(ch_handle* h, const float* code, float* out11, int B, ch_stream_t stream);
int  ch_color_predict(ch_handle* h, const float* code, float* out4, int B, ch_stream_t stream);

/* ---- shape branch (256x256 only: the Linear sizes fix it, shape_branch/model.py:85-89,120-122) ---------------
 * ch_shape_encode replaces mask_label_to_one_hot + split_hair_face + forward_hair_encoder(testing=True) +
 *   forward_face_encoder (ui/backend.py:81-86; shape_util.py:6-26; model.py:96-108,164-173):
 *   labels uint8 [B,256,256] (255 = no class) -> hair_code [B,16] (VAE mean), face_code [B,1024]; either output
 *   may be NULL to skip that encoder.
 * ch_shape_decode replaces forward_decode_by_code / forward_hair_decoder / forward_face_decoder / forward_decoder
 *   + mask_one_hot_to_label (model.py:175-199; shape_util.py:17-20; ui/backend.py:87-90,304-315):
 *   any of hair_logit [B,1,256,256], face_logit [B,18,256,256], labels uint8 [B,256,256], probs [B,19,256,256]
 *   may be NULL; hair_code NULL = face decoder only (ui/backend.py:420).
 * ch_shape_combine replaces Generator.forward_decoder on caller-made logits (ui/backend.py:421-424). */
int  ch_shape_encode(ch_handle* h, const uint8_t* labels, float* hair_code, float* face_code, int B, ch_stream_t stream);
int  ch_shape_decode(ch_handle* h, const float* hair_code, const float* face_code, float* hair_logit, float* face_logit,
                     uint8_t* labels, float* probs, int B, ch_stream_t stream);
int  ch_shape_combine(ch_handle* h, const float* hair_logit, const float* face_logit, uint8_t* labels, float* probs,
                      int B, ch_stream_t stream);

/* ---- BiSeNet face parser ---------------------------------------------------------------------------------------
 * Replaces FaceParsing.parsing_img's network part + swap_parsing_label_to_celeba_mask
 * (my_parsing_util.py:37-54; model.py:241-254 output [0] only; resnet.py:71-80):
 *   img float [B,3,H,W], already ImageNet-normalised (my_parsing_util.py:25-28); H, W multiples of 32
 *   labels uint8 [B,H,W] in CelebAMask-HQ ids;  logits (optional) float [B,19,H,W] = bilinear(align_corners) logits
 * The reference runs this network on CPU (the .cuda() calls are commented out, :37,41). */
int  ch_bisenet_parse(ch_handle* h, const float* img, uint8_t* labels, float* logits, int B, int H, int W,
                      ch_stream_t stream);

/* ---- Blending after the generator (the step that follows the hot path when Backend(blending=True)) ------------------
 * ch_blend_mask replaces hair_editor.py:297-305: hair = (target_parsing == 13) | (face_parsing == 13); out = cv2.dilate of
 *   hair with the 13x13 MORPH_ELLIPSE element, except on the target's background (label 0) where the 5x5 element is used.
 *   target_parsing, face_parsing, out: uint8 [H,W] device pointers (CelebAMask-HQ ids in, 0/1 out).
 * ch_poisson_blend replaces poisson_blending.poisson_blending (poisson_blending.py:29-87): same linear system (5-point
 *   Laplacian incl. the reference's border rows, identity rows for interior pixels with mask == 0), same gamma-2.2 round
 *   trip and uint8 truncation, solved matrix-free by conjugate gradients (Chronopoulos-Gear form, f64) instead of three sparse direct solves.
 *   source, target, out: uint8 [H,W,3] (cv2 layout); mask uint8 [H,W], non-zero = keep the SOURCE gradients (solve), zero =
 *   keep the target pixel; H, W >= 3.  Stops when ||r|| <= rel_tol * ||r0|| per channel or after max_iters iterations
 *   (recommended 1e-7 / 4000); *iters (host pointer, optional) receives the iteration count, NEGATED (INT_MIN for zero
 *   iterations) when the solve stopped at max_iters without reaching rel_tol -- checked once more after the last update
 *   (the reference uses a direct solve: an unconverged image is not its output).  Output agrees with the
 *   reference to +-1 grey level (the floor() after the gamma power amplifies last-bit differences of pow() and of the
 *   solve wherever the result sits on an integer boundary, e.g. every kept target pixel).  Run-to-run deterministic. */
int  ch_blend_mask(ch_handle* h, const uint8_t* target_parsing, const uint8_t* face_parsing, uint8_t* out, int H, int W,
                   ch_stream_t stream);
int  ch_poisson_blend(ch_handle* h, const uint8_t* source, const uint8_t* target, const uint8_t* mask, uint8_t* out, int H,
                      int W, int with_gamma, int max_iters, double rel_tol, int* iters, ch_stream_t stream);

/* Test hook: after the next ch_sean_generate calls, the activation produced at stage `name` ("fc", "<block>",
 * "<block>.ace_0" = tensor before leaky_relu, "<block>.conv_0", "<block>.shortcut") is also copied
 * (device-to-device, same stream) to `dev_ptr` (caller-sized: [B,C,r,r] floats).  dev_ptr NULL removes the tap. */
int  ch_sean_set_tap(ch_handle* h, const char* name, float* dev_ptr);

/* Diagnostic hook of the f16x3 / f16 paths (ctrlhair_amd/csrc/sh16.h): data-dependent activations are stored as f16 hi/lo
 * pairs with a power-of-two scale; each producer records the maximum of |value * 8| over its tensor and rewrites the
 * tensor with a corrected scale when that maximum left the window [0.5, 65504].  Copies to host_out[0..n) the maxima
 * recorded by the last ch_sean_generate batch chunk: entry 2i = output of ACE layer i (execution order), 2i+1 = its style
 * projections; 0 = not written.  Synchronises the device (not for the hot path). */
int  ch_sean_scale_report(ch_handle* h, float* host_out, int n);

/* Profiling hook (tools/ only): copies the first `bytes` of the generator's split-K scratch to host memory; with option
 * "sean.dbg" bit 256 the wave-specialised conv kernel leaves per-tile cycle stamps there.  Synchronises the device. */
int  ch_sean_debug_read(ch_handle* h, void* host_out, size_t bytes);

/* Roofline helper: the matrix-core issue rate this device sustains on an MFMA-only loop with non-trivial operands (no memory
 * traffic; ~ms_target ms; synchronises the device).  kind 0 = v_mfma_f32_32x32x2_f32, 1 = v_mfma_f32_32x32x16_f16.  Reported by
 * bench.py next to the spec peak (the spec figure assumes the 2.4 GHz boost clock). */
int  ch_mfma_peak(ch_handle* h, int kind, int ms_target, double* tflops);

/* Kernel-level timing hook for bench.py / roofline: when enabled, ch_sean_generate brackets every MFMA conv launch
 * with hipEvents on `stream`.  ch_profile_read synchronises those events and returns, for launches of `kind`
 * (0 = plain conv, 1 = SPADE conv with fused ACE epilogue, 2 = style-LUT GEMM, 3 = interior pass of a sparse ACE, <0 = all), their count, summed
 * duration (ms) and summed algorithmic flops / bytes.  A read with kind < 0 also clears the records. */
int  ch_profile_enable(ch_handle* h, int on);
int  ch_profile_read(ch_handle* h, int kind, int* launches, double* total_ms, double* flops, double* bytes);
/* As ch_profile_read; additionally `flops_executed`: the FLOPs the matrix cores actually ran.  They differ from `flops`
 * (the dense evaluation of every layer) for ACE launches served by the exact SPADE-interior reduction (option "sean.sparse",
 * ctrlhair_amd/csrc/ace_sparse.h): gamma/beta of SPADE.forward (/root/reference/sean_codes/models/networks/
 * normalization.py:249-257) depend on the 5x5 label neighbourhood only, so pixels whose neighbourhood is uniform take
 * per-label constants and only the compacted boundary pixels go through the conv. */
int  ch_profile_read_ex(ch_handle* h, int kind, int* launches, double* total_ms, double* flops, double* flops_executed,
                        double* bytes);

#ifdef __cplusplus
}
#endif
#endif
