// ch_api.cpp -- the C ABI of libctrlhair_hip.so (see include/ctrlhair_hip.h for the contract).
#include "../../include/ctrlhair_hip.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "aux_models.h"
#include "sean_model.h"
#include "kernels.h"

struct ch_handle {
    int device = 0;
    std::string err;
    chk::TensorStore tensors[4];
    chk::SeanModel sean;
    bool sean_ready = false;
    chk::ShapeModel shape;
    chk::ColorModel color;
    chk::BiSeNetModel bisenet;
    void* blend_ws = nullptr;        // Poisson CG workspace, grown on demand
    size_t blend_ws_bytes = 0;
};

namespace {
int fail(ch_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}
// Every entry point runs with the handle's device current and restores the caller's device on return (a process may own
// handles on several GPUs, or keep torch's current device elsewhere).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
}  // namespace

extern "C" {

int ch_abi_version(void) { return CH_ABI_VERSION; }

int ch_create(int device, ch_handle** out) {
    if (!out) return CH_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return CH_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CH_ERR_HIP;
    ch_handle* h = new (std::nothrow) ch_handle();
    if (!h) return CH_ERR_HIP;
    h->device = device;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        // kernels are built for gfx950 only; refuse instead of failing at the first launch
        h->err = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only";
        *out = h;
        return CH_ERR_HIP;
    }
    *out = h;
    return CH_OK;
}

void ch_destroy(ch_handle* h) {
    if (!h) return;
    {
        DeviceGuard g(h->device);
        (void)hipDeviceSynchronize();
        h->sean.destroy();
        h->shape.destroy();
        h->color.destroy();
        h->bisenet.destroy();
        if (h->blend_ws) (void)hipFree(h->blend_ws);
        h->blend_ws = nullptr;
    }
    delete h;
}

const char* ch_last_error(const ch_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ch_load_tensor(ch_handle* h, int model, const char* name, const void* host, int dtype, const int64_t* shape,
                   int ndim) {
    if (!h || !name || !host || ndim < 0 || ndim > 8 || (ndim > 0 && !shape)) return fail(h, CH_ERR_ARG, "bad argument");
    if (model < 0 || model > CH_MODEL_BISENET) return fail(h, CH_ERR_ARG, "unknown model id");
    if (dtype != CH_F32 && dtype != CH_I64) return fail(h, CH_ERR_ARG, "unsupported dtype");
    try {
        chk::HostTensor t;
        t.dtype = dtype;
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) {
            if (shape[i] < 0) return fail(h, CH_ERR_ARG, "negative dimension");
            t.shape.push_back(shape[i]);
            n *= (size_t)shape[i];
        }
        const size_t bytes = n * (dtype == CH_F32 ? 4 : 8);
        t.data.assign(static_cast<const char*>(host), static_cast<const char*>(host) + bytes);
        h->tensors[model][name] = std::move(t);
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_WEIGHTS, std::string("ch_load_tensor: ") + e.what());
    }
    return CH_OK;
}

int ch_finalize(ch_handle* h, int model, int max_batch, int max_size) {
    if (!h) return CH_ERR_ARG;
    if (model < 0 || model > CH_MODEL_BISENET) return fail(h, CH_ERR_ARG, "unknown model id");
    DeviceGuard guard(h->device);
    if (model != CH_MODEL_SEAN) {
        try {
            std::string e;
            if (model == CH_MODEL_SHAPE) { h->shape.destroy(); e = h->shape.build(h->tensors[model], max_batch); if (!e.empty()) h->shape.destroy(); }
            if (model == CH_MODEL_COLOR) { h->color.destroy(); e = h->color.build(h->tensors[model], max_batch); if (!e.empty()) h->color.destroy(); }
            if (model == CH_MODEL_BISENET) { h->bisenet.destroy(); e = h->bisenet.build(h->tensors[model], max_batch, max_size); if (!e.empty()) h->bisenet.destroy(); }
            if (!e.empty()) return fail(h, CH_ERR_WEIGHTS, "ch_finalize: " + e);
            h->tensors[model].clear();
        } catch (const std::exception& e) {
            return fail(h, CH_ERR_WEIGHTS, std::string("ch_finalize: ") + e.what());
        }
        return CH_OK;
    }
    try {
        h->sean.destroy();
        h->sean_ready = false;
        std::string e = h->sean.build(h->tensors[model], max_batch, max_size);
        if (!e.empty()) {
            h->sean.destroy();
            return fail(h, CH_ERR_WEIGHTS, "ch_finalize: " + e);
        }
        h->tensors[model].clear();   // host copies no longer needed
        h->sean_ready = true;
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_WEIGHTS, std::string("ch_finalize: ") + e.what());
    }
    return CH_OK;
}

size_t ch_sean_noise_floats(const ch_handle* h, int S) {
    if (!h || !h->sean_ready || S <= 0) return 0;
    return h->sean.noise_floats(S);
}

int ch_sean_generate(ch_handle* h, const uint8_t* labels, const float* codes, const float* noise, uint64_t seed,
                     float* out, int B, int S, ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!h->sean_ready) return fail(h, CH_ERR_STATE, "ch_sean_generate: SEAN weights not finalized");
    if (!labels || !codes || !out) return fail(h, CH_ERR_ARG, "ch_sean_generate: null pointer");
    DeviceGuard guard(h->device);
    try {
        std::string e = h->sean.generate(labels, codes, noise, seed, out, B, S, static_cast<hipStream_t>(stream));
        if (!e.empty()) return fail(h, CH_ERR_HIP, "ch_sean_generate: " + e);
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_HIP, std::string("ch_sean_generate: ") + e.what());
    }
    return CH_OK;
}

int ch_sean_draw_noise(ch_handle* h, uint64_t seed, float* noise, int B, int S, ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!h->sean_ready) return fail(h, CH_ERR_STATE, "ch_sean_draw_noise: SEAN weights not finalized");
    if (!noise || B < 1) return fail(h, CH_ERR_ARG, "ch_sean_draw_noise: bad argument");
    DeviceGuard guard(h->device);
    std::string e = h->sean.draw_noise(seed, noise, B, S, static_cast<hipStream_t>(stream));
    if (!e.empty()) return fail(h, CH_ERR_HIP, "ch_sean_draw_noise: " + e);
    return CH_OK;
}

int ch_sean_encode(ch_handle* h, const float* img, const uint8_t* labels, float* codes, int B, int S,
                   ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!h->sean_ready) return fail(h, CH_ERR_STATE, "ch_sean_encode: SEAN weights not finalized");
    if (!img || !labels || !codes || B < 1) return fail(h, CH_ERR_ARG, "ch_sean_encode: bad argument");
    DeviceGuard guard(h->device);
    try {
        std::string e = h->sean.encode(img, labels, codes, B, S, static_cast<hipStream_t>(stream));
        if (!e.empty()) return fail(h, CH_ERR_HIP, "ch_sean_encode: " + e);
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_HIP, std::string("ch_sean_encode: ") + e.what());
    }
    return CH_OK;
}

int ch_sean_encode_features(ch_handle* h, const float* img, int B, int S, ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!h->sean_ready) return fail(h, CH_ERR_STATE, "ch_sean_encode_features: SEAN weights not finalized");
    if (!img || B < 1) return fail(h, CH_ERR_ARG, "ch_sean_encode_features: bad argument");
    DeviceGuard guard(h->device);
    try {
        std::string e = h->sean.encode(img, nullptr, nullptr, B, S, static_cast<hipStream_t>(stream), 1);
        if (!e.empty()) return fail(h, CH_ERR_HIP, "ch_sean_encode_features: " + e);
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_HIP, std::string("ch_sean_encode_features: ") + e.what());
    }
    return CH_OK;
}
int ch_sean_encode_regions(ch_handle* h, const uint8_t* labels, float* codes, int B, int S, ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!h->sean_ready) return fail(h, CH_ERR_STATE, "ch_sean_encode_regions: SEAN weights not finalized");
    if (!labels || !codes || B < 1) return fail(h, CH_ERR_ARG, "ch_sean_encode_regions: bad argument");
    DeviceGuard guard(h->device);
    try {
        std::string e = h->sean.encode(nullptr, labels, codes, B, S, static_cast<hipStream_t>(stream), 2);
        if (!e.empty()) return fail(h, CH_ERR_HIP, "ch_sean_encode_regions: " + e);
    } catch (const std::exception& e) {
        return fail(h, CH_ERR_HIP, std::string("ch_sean_encode_regions: ") + e.what());
    }
    return CH_OK;
}

#define CH_CALL(name, expr)                                                             \
    if (!h) return CH_ERR_ARG;                                                          \
    DeviceGuard guard(h->device);                                                       \
    try {                                                                               \
        std::string e = (expr);                                                         \
        if (!e.empty()) return fail(h, e.find("not finalized") != std::string::npos ? CH_ERR_STATE : CH_ERR_HIP, \
                                    std::string(name) + ": " + e);                      \
    } catch (const std::exception& ex) {                                                \
        return fail(h, CH_ERR_HIP, std::string(name) + ": " + ex.what());               \
    }                                                                                   \
    return CH_OK;

int ch_color_generate(ch_handle* h, const float* noise, const float* cond, float* code, int B, ch_stream_t stream) {
    if (h && (!noise || !cond || !code || B < 1)) return fail(h, CH_ERR_ARG, "ch_color_generate: bad argument");
    CH_CALL("ch_color_generate", h->color.generate(noise, cond, code, B, static_cast<hipStream_t>(stream)))
}
int ch_color_encode(ch_handle* h, const float* code, float* out11, int B, ch_stream_t stream) {
    if (h && (!code || !out11 || B < 1)) return fail(h, CH_ERR_ARG, "ch_color_encode: bad argument");
    CH_CALL("ch_color_encode", h->color.encode(code, out11, B, static_cast<hipStream_t>(stream)))
}
int ch_color_predict(ch_handle* h, const float* code, float* out4, int B, ch_stream_t stream) {
    if (h && (!code || !out4 || B < 1)) return fail(h, CH_ERR_ARG, "ch_color_predict: bad argument");
    CH_CALL("ch_color_predict", h->color.predict(code, out4, B, static_cast<hipStream_t>(stream)))
}
int ch_shape_encode(ch_handle* h, const uint8_t* labels, float* hair_code, float* face_code, int B, ch_stream_t stream) {
    if (h && (!labels || (!hair_code && !face_code) || B < 1)) return fail(h, CH_ERR_ARG, "ch_shape_encode: bad argument");
    CH_CALL("ch_shape_encode", h->shape.encode(labels, hair_code, face_code, B, static_cast<hipStream_t>(stream)))
}
int ch_shape_decode(ch_handle* h, const float* hair_code, const float* face_code, float* hair_logit, float* face_logit,
                    uint8_t* labels, float* probs, int B, ch_stream_t stream) {
    if (h && (!face_code || B < 1 || ((labels || probs || hair_logit) && !hair_code)))
        return fail(h, CH_ERR_ARG, "ch_shape_decode: bad argument");
    CH_CALL("ch_shape_decode", h->shape.decode(hair_code, face_code, hair_logit, face_logit, labels, probs, B,
                                               static_cast<hipStream_t>(stream)))
}
int ch_shape_combine(ch_handle* h, const float* hair_logit, const float* face_logit, uint8_t* labels, float* probs, int B,
                     ch_stream_t stream) {
    if (h && (!hair_logit || !face_logit || !labels || B < 1)) return fail(h, CH_ERR_ARG, "ch_shape_combine: bad argument");
    CH_CALL("ch_shape_combine", h->shape.combine(hair_logit, face_logit, labels, probs, B, static_cast<hipStream_t>(stream)))
}
int ch_bisenet_parse(ch_handle* h, const float* img, uint8_t* labels, float* logits, int B, int H, int W,
                     ch_stream_t stream) {
    if (h && (!img || !labels || B < 1)) return fail(h, CH_ERR_ARG, "ch_bisenet_parse: bad argument");
    CH_CALL("ch_bisenet_parse", h->bisenet.parse(img, labels, logits, B, H, W, static_cast<hipStream_t>(stream)))
}

int ch_blend_mask(ch_handle* h, const uint8_t* target_parsing, const uint8_t* face_parsing, uint8_t* out, int H, int W,
                  ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!target_parsing || !face_parsing || !out || H < 1 || W < 1) return fail(h, CH_ERR_ARG, "ch_blend_mask: bad argument");
    DeviceGuard guard(h->device);
    hipError_t e = chk::blend_mask(target_parsing, face_parsing, out, H, W, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CH_OK : fail(h, CH_ERR_HIP, std::string("ch_blend_mask: ") + hipGetErrorString(e));
}

int ch_poisson_blend(ch_handle* h, const uint8_t* source, const uint8_t* target, const uint8_t* mask, uint8_t* out, int H, int W,
                     int with_gamma, int max_iters, double rel_tol, int* iters, ch_stream_t stream) {
    if (!h) return CH_ERR_ARG;
    if (!source || !target || !mask || !out || H < 3 || W < 3 || max_iters < 0 || !(rel_tol >= 0.0))
        return fail(h, CH_ERR_ARG, "ch_poisson_blend: bad argument (images need H, W >= 3)");
    DeviceGuard guard(h->device);
    const size_t need = chk::poisson_workspace_bytes(H, W);
    if (need > h->blend_ws_bytes) {
        if (h->blend_ws) (void)hipFree(h->blend_ws);
        h->blend_ws = nullptr;
        h->blend_ws_bytes = 0;
        if (hipMalloc(&h->blend_ws, need) != hipSuccess) return fail(h, CH_ERR_HIP, "ch_poisson_blend: workspace allocation failed");
        h->blend_ws_bytes = need;
    }
    hipError_t e = chk::poisson_blend(source, target, mask, out, H, W, with_gamma, max_iters, rel_tol, h->blend_ws, iters,
                                      static_cast<hipStream_t>(stream));
    return e == hipSuccess ? CH_OK : fail(h, CH_ERR_HIP, std::string("ch_poisson_blend: ") + hipGetErrorString(e));
}

int ch_sean_set_tap(ch_handle* h, const char* name, float* dev_ptr) {
    if (!h || !name) return CH_ERR_ARG;
    if (dev_ptr) h->sean.taps[name] = dev_ptr;
    else h->sean.taps.erase(name);
    return CH_OK;
}

int ch_sean_scale_report(ch_handle* h, float* host_out, int n) {
    if (!h || !host_out || n < 0) return CH_ERR_ARG;
    if (!h->sean_ready || !h->sean.amax_slots) return fail(h, CH_ERR_STATE, "ch_sean_scale_report: SEAN weights not finalized");
    DeviceGuard guard(h->device);
    unsigned raw[64];
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(raw, h->sean.amax_slots, sizeof raw, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(h, CH_ERR_HIP, "ch_sean_scale_report: device read failed");
    for (int i = 0; i < n; ++i) {
        float v = 0.f;
        if (i < 64) std::memcpy(&v, &raw[i], 4);
        host_out[i] = v;
    }
    return CH_OK;
}

int ch_sean_debug_read(ch_handle* h, void* host_out, size_t bytes) {
    if (!h || !host_out) return CH_ERR_ARG;
    if (!h->sean_ready || !h->sean.splitk_ws) return fail(h, CH_ERR_STATE, "ch_sean_debug_read: SEAN weights not finalized");
    DeviceGuard guard(h->device);
    if ((long long)bytes > h->sean.splitk_cap * 4) return fail(h, CH_ERR_ARG, "ch_sean_debug_read: too many bytes");
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, h->sean.splitk_ws, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(h, CH_ERR_HIP, "ch_sean_debug_read: device read failed");
    return CH_OK;
}

int ch_set_option(ch_handle* h, const char* key, int value) {
    if (!h || !key) return CH_ERR_ARG;
    if (std::strcmp(key, "sean.f16x3") == 0) {
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.f16x3) must precede ch_finalize");
        // 0 exact f32 | 1 f16x3 split operands (f32-class) | 2 single-term f16 operands | 3 single-term bf16 operands
        h->sean.use_sh16 = value != 0;
        h->sean.terms = value == 2 ? 1 : (value == 3 ? 2 : 3);
        return CH_OK;
    }
    if (std::strcmp(key, "shape.f16x3") == 0) {
        if (h->shape.ready) return fail(h, CH_ERR_STATE, "ch_set_option(shape.f16x3) must precede ch_finalize");
        h->shape.use_sh16 = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "shape.overlap") == 0) {     // 1 = the hair encoder / decoder on a side stream beside the face one (default), 0 = one after the other
        if (h->shape.ready) return fail(h, CH_ERR_STATE, "ch_set_option(shape.overlap) must precede ch_finalize");
        h->shape.overlap = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "shape.enc_lut") == 0) {     // exact-f32 shape encoders: 1 = layer 0 as a label table (default), 0 = through the conv kernel
        if (h->shape.ready) return fail(h, CH_ERR_STATE, "ch_set_option(shape.enc_lut) must precede ch_finalize");
        h->shape.enc_l0_lut = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "bisenet.f16x3") == 0) {
        if (h->bisenet.ready) return fail(h, CH_ERR_STATE, "ch_set_option(bisenet.f16x3) must precede ch_finalize");
        h->bisenet.use_sh16 = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "aux.wino") == 0) {        // exact-f32 kernels of the shape VAE / BiSeNet: 3x3 stride-1 convs as Winograd F(2x2,3x3) (default 1)
        h->shape.wino = h->bisenet.wino = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.ahead") == 0) {      // run-ahead mode up to `value` images of 512x512 per chunk (0 = off)
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.ahead) must precede ch_finalize");
        h->sean.ahead_pixels = (long long)value * 512 * 512;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino") == 0) {       // exact-f32 path: 3x3 convs as Winograd F(2x2,3x3) on the f32 matrix cores (conv_wino.h)
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.wino) must precede ch_finalize");
        h->sean.wino = value < 0 ? 0 : (value > 2 ? 2 : value);      // 1 = F(2x2,3x3); 2 = + the ResBlock convs as F(4x4,3x3) (default)
        return CH_OK;
    }
    if (std::strcmp(key, "sean.lut_grouped") == 0) {    // exact-f32 path: 1 = the style LUTs of a chunk from one grouped GEMM launch (default)
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.lut_grouped) must precede ch_finalize");
        h->sean.lut_grouped = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.overlap") == 0) {      // CUs of the side streams of the overlap mode (sean_model.h), 0 = off
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.overlap) must precede ch_finalize");
        h->sean.overlap = value < 0 ? 0 : value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino4_force") == 0) {  // 1 = F(4x4,3x3) wherever the shape allows, even with fewer tasks than CUs (tests / measurements)
        h->sean.wino4_force = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.patch") == 0) {        // 1 = pre-gathered hidden-activation patches for levels with few boundary quads (default), 0 = planes only
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.patch) must precede ch_finalize");
        h->sean.patch = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.convt_gemm") == 0) {   // exact-f32 Zencoder: 1 = the ConvTranspose as four phase GEMMs over shifted views (default), 0 = four Winograd phase convs
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.convt_gemm) must precede ch_finalize");
        h->sean.convt_gemm = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.edge") == 0) {         // 1 = straight-edge pixels from per-code table rows in the interior pass (default), 0 = through the boundary conv
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.edge) must precede ch_finalize");
        h->sean.edge = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.batch_invariant") == 0) {      // exact-f32 path: 1 = kernel choices independent of the batch size of a call (default 0)
        h->sean.batch_inv = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino4v") == 0) {       // 1 = pre-transformed-input route of the F(4x4,3x3) layers with many GEMM rows (conv_wino4v.h; default), 0 = off
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.wino4v) must precede ch_finalize");
        h->sean.wino4v = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino4_ace") == 0) {    // largest level (pixels) whose SPADE convs run as F(4x4,3x3) over every tile; 0 = none
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.wino4_ace) must precede ch_finalize");
        h->sean.wino4_ace_max_r = value < 0 ? 0 : value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.hidden_wq") == 0) {    // Winograd ACE path: 1 = hidden activations + one-hot planes from spade_hidden_wq (default)
        h->sean.hidden_wq = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino_gather") == 0) {    // 1 = gather mode of the Winograd ACE kernel (default), 0 = tile mode
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.wino_gather) must precede ch_finalize");
        h->sean.wino_gather = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.wino_th") == 0) {    // tile height 16 / 32 of the Winograd ACE kernel (0 = chosen per level)
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.wino_th) must precede ch_finalize");
        h->sean.wino_th = value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.sparse") == 0) {     // exact SPADE-interior reduction (ace_sparse.h); buffers are sized at ch_finalize
        // after ch_finalize the reduction can be switched OFF (and back on); a handle finalised without it has no classification
        // buffers, and turning it on would silently do nothing
        if (h->sean_ready && value != 0 && !h->sean.gtab)
            return fail(h, CH_ERR_STATE, "ch_set_option(sean.sparse = 1): the handle was finalised with sean.sparse = 0 (no buffers); set it before ch_finalize");
        h->sean.sparse = value != 0;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.sh16_compact") == 0) {   // f16x3 path: pixel-level compaction (1) or tile skipping only (0)
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.sh16_compact) must precede ch_finalize");
        h->sean.sh16_compact = value < 0 ? 0 : (value > 2 ? 2 : value);    // 2: without pair entries (A/B)
        return CH_OK;
    }
    if (std::strcmp(key, "sean.sparse_th") == 0) {  // tile height of the compaction: 8 / 16, 0 = chosen per layer
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.sparse_th) must precede ch_finalize");
        h->sean.sparse_th = value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.sparse_min") == 0) { // smallest ACE resolution served by the sparse path
        if (h->sean_ready) return fail(h, CH_ERR_STATE, "ch_set_option(sean.sparse_min) must precede ch_finalize");
        h->sean.sparse_min_r = value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.dbg_sel") == 0) {
        h->sean.dbg_sel = value;
        return CH_OK;
    }
    if (std::strcmp(key, "sean.dbg") == 0) {
#ifndef CH_ABLATE
        if (value & CH_ABLATE_DBG_MASK)
            return fail(h, CH_ERR_ARG, "ch_set_option(sean.dbg): timing-ablation / superseded-kernel bits need a library built with -DCH_ABLATE "
                                       "(make -C ctrlhair_amd/csrc ABLATE=1)");
#endif
        h->sean.dbg = value;
        return CH_OK;
    }
    return fail(h, CH_ERR_ARG, std::string("unknown option '") + key + "'");
}

int ch_mfma_peak(ch_handle* h, int kind, int ms_target, double* tflops) {
    if (!h || !tflops || kind < 0 || kind > 1 || ms_target < 1 || ms_target > 2000) return fail(h, CH_ERR_ARG, "ch_mfma_peak: bad argument");
    DeviceGuard guard(h->device);
    hipError_t e = chk::mfma_peak(kind, ms_target, tflops, nullptr);
    return e == hipSuccess ? CH_OK : fail(h, CH_ERR_HIP, std::string("ch_mfma_peak: ") + hipGetErrorString(e));
}

int ch_profile_enable(ch_handle* h, int on) {
    if (!h) return CH_ERR_ARG;
    h->sean.prof_on = on != 0;
    return CH_OK;
}

int ch_profile_read(ch_handle* h, int kind, int* launches, double* total_ms, double* flops, double* bytes) {
    return ch_profile_read_ex(h, kind, launches, total_ms, flops, nullptr, bytes);
}

int ch_profile_read_ex(ch_handle* h, int kind, int* launches, double* total_ms, double* flops, double* flops_executed,
                       double* bytes) {
    if (!h) return CH_ERR_ARG;
    DeviceGuard guard(h->device);
    int n = 0;
    double ms = 0, fl = 0, fx = 0, by = 0;
    for (auto& r : h->sean.prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return fail(h, CH_ERR_HIP, "hipEventSynchronize failed");
        if (kind < 0 || r.kind == kind) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return fail(h, CH_ERR_HIP, "hipEventElapsedTime failed");
            ms += t;
            fl += r.flops;
            if (r.sp_stat) {        // sparse ACE launch: the matrix cores ran over the compacted boundary sub-tiles only
                int st[4] = {0, 0, 0, 0};
                if (hipMemcpy(st, r.sp_stat, sizeof st, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail(h, CH_ERR_HIP, "ch_profile_read: work-list statistics read failed");
                fx += (double)st[3] * r.sp_flops_unit;
                by += r.sp_bytes_fixed + r.sp_bytes_px * (r.kind == 3 ? r.sp_npix - (double)st[1] : (double)st[1]);
            } else {
                fx += r.flops_exec >= 0.0 ? r.flops_exec : r.flops;
                by += r.bytes;
            }
            ++n;
        }
    }
    if (flops_executed) *flops_executed = fx;
    if (kind < 0 || true) {
        // records are consumed only by a read with kind < 0 (read specific kinds first)
        if (kind < 0) {
            for (auto& r : h->sean.prof) {
                h->sean.ev_pool.push_back(r.e0);
                h->sean.ev_pool.push_back(r.e1);
            }
            h->sean.prof.clear();
            h->sean.prof_stats_used = 0;
        }
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return CH_OK;
}

}  // extern "C"
