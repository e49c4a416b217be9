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
    CH_MODEL_SEAN = 0          /* sean_codes/models/networks/generator.py: SPADEGenerator (+Zencoder) */
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

/* Fold + pack + upload the loaded tensors: spectral-norm sigma (torch spectral_norm eval semantics,
 * architecture.py:42-46), eval-BN running stats -> per-channel affine (sync_batchnorm/batchnorm.py:52-55),
 * sigmoid(blending) folded into the SPADE / style weights (normalization.py:177-181), one-hot convs -> label
 * LUTs, MFMA operand layouts.  Sizes the workspace arena for images up to max_size x max_size in chunks of
 * max_batch.  ngf is inferred from the tensors.  Synchronises the device. */
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

/* Replaces Pix2PixModel.forward(data, mode='style_code') -> Zencoder.forward
 * (pix2pix_model.py:69-72; architecture.py:177-207; callers hair_editor.py:149-157 get_code, :208-231):
 * conv stack (reflection-padded 3x3, two stride-2 convs, ConvTranspose2d, InstanceNorm + LeakyReLU, tanh) at
 * S/2 resolution, then per-region average pooling with the label map nearest-down-sampled to S/2.
 *   img    : device float [B,3,S,S] in [-1,1]        labels : device uint8 [B,S,S]
 *   codes  : device float [B,19,512] (rows of absent regions are 0)
 * Requires the Zencoder.* tensors to have been loaded before ch_finalize. */
int  ch_sean_encode(ch_handle* h, const float* img, const uint8_t* labels, float* codes, int B, int S,
                    ch_stream_t stream);

/* Test hook: after the next ch_sean_generate calls, the activation produced at stage `name` ("fc", "<block>",
 * "<block>.ace_0" = tensor before leaky_relu, "<block>.conv_0", "<block>.shortcut") is also copied
 * (device-to-device, same stream) to `dev_ptr` (caller-sized: [B,C,r,r] floats).  dev_ptr NULL removes the tap. */
int  ch_sean_set_tap(ch_handle* h, const char* name, float* dev_ptr);

/* Kernel-level timing hook for bench.py / roofline: when enabled, ch_sean_generate brackets every MFMA conv launch
 * with hipEvents on `stream`.  ch_profile_read synchronises those events and returns, for launches of `kind`
 * (0 = plain conv, 1 = SPADE conv with fused ACE epilogue, 2 = style-LUT GEMM, <0 = all), their count, summed
 * duration (ms) and summed algorithmic flops / bytes.  A read with kind < 0 also clears the records. */
int  ch_profile_enable(ch_handle* h, int on);
int  ch_profile_read(ch_handle* h, int kind, int* launches, double* total_ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif
