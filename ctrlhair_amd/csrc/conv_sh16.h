// conv_sh16.h -- fp32-class implicit-GEMM 3x3 / 1x1 convolution on the *f16* matrix cores of gfx950
// (v_mfma_f32_32x32x16_f16, 16x the f32-MFMA rate) by the 3-term split-operand identity
//
//      a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi ,   x_hi = f16(x),  x_lo = f16(x - x_hi)
//
// x_hi + x_lo carries 22 mantissa bits; the dropped a_lo*b_lo term is < 2^-22 relative, products are exact in the
// f32 accumulator, so the result is f32-class (measured vs the exact-f32 MFMA kernel: see DESIGN.md) at 3 MFMAs of
// 32 cycles per 16-deep k-step instead of 8 MFMAs of 64 cycles: 5.3x the f32-MFMA throughput ceiling.
//
// Activations feeding these convs live in HBM in the "SH16" layout, designed for the MFMA B fragment:
//      [B][C/8][2 (hi|lo)][H][W][8 channels] of _Float16    (16-byte unit = 8 channels of one pixel, hi or lo plane)
// = 4 bytes per element like f32, but a lane's B fragment (8 consecutive k for its pixel) is ONE aligned 16-byte unit,
// pixels of a row are contiguous units (coalesced staging, conflict-free ds_read_b128).  Producers are our own kernels
// (ACE epilogue below, onehot_conv3x3_sh16), so the split costs no extra pass.
// Weights are pre-split and packed on the host into per-lane A fragments:
//      [wave tile 64 rows][chunk 16 ch][tap][M-subtile][hi|lo][lane][8 halfs]
//
// Block = 256 threads = 4 waves, all on the same 64 GEMM rows (A fragments are shared through L1), each wave 128 px:
// block tile 64 rows x 512 px; 8 f32 accumulators (32x32) per wave; 24 MFMAs per k-step per wave.
#pragma once
#include <type_traits>

#include "conv_mfma.h"
#include "sh16.h"

namespace chk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int KS, int TW, int TH, int TB>
struct ShCfg {
    static constexpr int CK = 16;                       // channels per chunk = one k-step per tap
    static constexpr int HALO = KS / 2;
    static constexpr int PW = TW + 2 * HALO, PH = TH + 2 * HALO;
    static constexpr int PLANE = TB * PH * PW;          // 16-byte units per (group, hi|lo) plane
    static constexpr int UNITS = 4 * PLANE;             // 2 groups x (hi, lo)
    static constexpr int NLOAD = (UNITS + 255) / 256;
    static constexpr int LDS_BYTES = 2 * UNITS * 16;    // double buffered
    static_assert(TW * TH * TB == 512, "block tile is 512 pixels");
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// TERMS: 3 = three-term f16 split (f32-class) | 1 = f16 operands (hi planes only) | 2 = bf16 operands (the hi planes hold
// bf16 bit patterns instead of f16; same layouts, same kernels, v_mfma_f32_32x32x16_bf16) -- BASELINE.json configs[4]
template <int TERMS>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
    if constexpr (TERMS == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// ACE modulation of 4 consecutive channels of one pixel, written on float2 pairs so that the arithmetic maps onto the
// packed-f32 VALU ops (v_pk_add/mul/fma_f32) and v_cvt_pk_f16_f32:
//      o = s_out * act((a*x + n*nz + d) * (1 + gamma) + beta),   act = max(o, slope*o)  (slope 1 / 0.2 / 0)
// The output scale s_out (sh16.h) costs nothing: the caller passes st = 2^-k / s_in * s_out (the accumulator scale, one
// scalar per wave tile) and bg = s_out * (1 + bias_g + style_g), bb = s_out * (bias_b + style_b), so that
// gamma' = acc_g * st + bg and beta' = acc_b * st + bb are already s_out * (1 + gamma) and s_out * beta.  RESC (second
// pass of a tensor whose maximum left the f16 window): one more multiply by the corrective power of two.  !RESC: the
// running max |o| is kept in `amax`.  f32 -> f16 conversions saturate (MODE.FP16_OVFL, set at kernel entry).
// Returns the f16 hi halves in .x/.y and the residual lo halves in .z/.w (two channels per dword).
template <bool RESC, bool BF = false>
__device__ __forceinline__ uint4 ace_quad(float g0, float g1, float g2, float g3, float b0, float b1, float b2, float b3,
                                          float st, const float4& bg, const float4& bb, const float4& pa, const float4& pd,
                                          const float4& pn, const float4& x4, float nz, float slope, float extra, float& amax) {
    uint4 w;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 ag = h ? f32x2{g2, g3} : f32x2{g0, g1}, ab = h ? f32x2{b2, b3} : f32x2{b0, b1};
        const f32x2 vbg = h ? f32x2{bg.z, bg.w} : f32x2{bg.x, bg.y}, vbb = h ? f32x2{bb.z, bb.w} : f32x2{bb.x, bb.y};
        const f32x2 va = h ? f32x2{pa.z, pa.w} : f32x2{pa.x, pa.y}, vd = h ? f32x2{pd.z, pd.w} : f32x2{pd.x, pd.y};
        const f32x2 vn = h ? f32x2{pn.z, pn.w} : f32x2{pn.x, pn.y}, vx = h ? f32x2{x4.z, x4.w} : f32x2{x4.x, x4.y};
        const f32x2 gam1 = ag * st + vbg;
        const f32x2 bet = ab * st + vbb;
        const f32x2 nrm = va * vx + (vn * nz + vd);
        f32x2 o = nrm * gam1 + bet;
        const f32x2 os = o * slope;
        o.x = fmaxf(o.x, os.x);
        o.y = fmaxf(o.y, os.y);
        if (RESC) o = o * extra;
        else amax = fmaxf(fmaxf(amax, fabsf(o.x)), fabsf(o.y));       // v_max3_f32 with |.| source modifiers
        if constexpr (BF) {                                  // bf16 operands: hi plane = bf16 bits, lo plane unused
            const bf16x2 hb = __builtin_convertvector(o, bf16x2);
            if (h == 0) { w.x = __builtin_bit_cast(unsigned, hb); w.z = 0u; }
            else        { w.y = __builtin_bit_cast(unsigned, hb); w.w = 0u; }
        } else {
            const f16x2 hh = __builtin_convertvector(o, f16x2);
            const f32x2 lo = o - __builtin_convertvector(hh, f32x2);
            const f16x2 ll = __builtin_convertvector(lo, f16x2);
            if (h == 0) { w.x = __builtin_bit_cast(unsigned, hh); w.z = __builtin_bit_cast(unsigned, ll); }
            else        { w.y = __builtin_bit_cast(unsigned, hh); w.w = __builtin_bit_cast(unsigned, ll); }
        }
    }
    return w;
}

// Scales in effect for an SH16 conv: returns 1 / s_in of the accumulators (first-pass scale x dynamic factor of the input's
// slot) and, for a fused second operand (conv_1 + conv_s on one accumulator), the powers of two `mul1` / `mul2` the two
// inputs' units are multiplied by while staged.  The weights of the two operands are packed so that, with both tensors at
// their first-pass scales, the products already share the accumulator scale (sean_model.cpp: the shortcut activations'
// first-pass scale carries the 2^D that aligns the two weight matrices).  If a second pass changed a scale, the operands
// are brought back together: in2 by e1 / e2, and both down by a common factor if that would push in2 out of the f16 range.
__device__ __forceinline__ float sh16_in_scale_inv(const ConvParams& p, float* mul1 = nullptr, float* mul2 = nullptr) {
    float isi = p.in_scale_inv != 0.f ? p.in_scale_inv : 1.f;
    float m1 = 1.f, m2 = 1.f;
    if (p.in_amax) {
        const float e1 = sh16_dyn_extra(*p.in_amax);
        if (p.in2_amax) {
            const unsigned a2b = *p.in2_amax;
            const float e2 = sh16_dyn_extra(a2b);
            float a2;
            __builtin_memcpy(&a2, &a2b, 4);
            const float t = a2 * e1;                         // max |in2 unit| after the e1 / e2 alignment
            if (t >= 32768.f && t < 3.0e38f) m1 = sh16_scale_for_bound(t);
            m2 = m1 * e1 / e2;
        }
        isi /= e1 * m1;
    }
    if (mul1) *mul1 = m1;
    if (mul2) *mul2 = m2;
    return isi * (p.out_mul != 0.f ? p.out_mul : 1.f);
}

__device__ __forceinline__ float act_slope(int act) { return act == ACT_NONE ? 1.f : (act == ACT_LRELU ? 0.2f : 0.f); }

// lanes l / l+32 hold channels 0-3 / 4-7 of the same pixel: one v_permlane32_swap per dword leaves lanes < 32 with the
// 8 hi halves and lanes >= 32 with the 8 lo halves, i.e. one 16-byte SH16 unit per lane
__device__ __forceinline__ uint4 sh16_pair_swap(const uint4& w) {
    auto r0 = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
    return make_uint4(r0[0], r1[0], r0[1], r1[1]);
}

// Pixel of slot `idx` of a block tile (idx = wave * 128 + sub-tile * 32 + lane % 32).  TW == 16: a 32-pixel sub-tile is two rows
// of 16 pixels whose patch rows start PW units apart in LDS.  ds_read_b128 serves a wave in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): with PW = 18 and the plain mapping, lanes 12,13 (row 0, units 12,13) and
// lanes 26,27 (row 1, units 18+10, 18+11 = 12, 13 mod 16) of one group fall on the same 16-byte bank slots -- a 2-way conflict
// in every group, the 44-53 % of the LDS-conflict column these tile variants showed (profiles/r02_pipeline_pmc.md).  Rotating
// the odd row by `rot` = (16 - PW % 16) % 16 pixels makes the 16 units of every group distinct mod 16; the epilogue uses the
// same mapping, so a lane still owns the pixel whose B fragment it fed (stores stay inside the same 64-byte row segment).
template <int TW, int TH>
__device__ __forceinline__ void sh16_slot_px(int idx, int rot, int& tx, int& ty, int& tb) {
    tx = idx % TW;
    ty = (idx / TW) % TH;
    tb = idx / (TW * TH);
    if constexpr (TW == 16) {
        if (ty & 1) tx = (tx + rot) & 15;
    }
}

// D2S (EPI_PLAIN): depth-to-space store.  GEMM row = phase * C + channel (C = Mrows / 4, phase = (py, px)); pixel (y, x) of the
// conv grid lands at (2y + py, 2x + px) of the C-channel, 2H x 2W output: ConvTranspose2d(k3, s2, p1, op1) as ONE 2x2-tap
// conv at the input resolution (each output phase only sees the taps that reach it: 9 of the 16 (tap, phase) weights are
// non-zero, against 9 of 36 positions for a 3x3 conv over the zero-inserted x2 view).  bias is indexed by channel.
template <int TW, int TH, int TB, int EPI, bool BF = false, bool D2S = false>
__device__ __forceinline__ void sh16_epilogue(const ConvParams& p, f32x16 (&acc)[2][4], int mtile64, int wn, int lane,
                                              int x0, int y0, int b0, int ks = 0, int rot = 0) {
    const int HW = p.H * p.W;
    // ---- epilogue ------------------------------------------------------------------------------------------
    // Per-channel parameters are loaded ONCE per wave as float4 (a lane's 16 rows are 4 runs of 4 consecutive
    // channels), not per pixel: the per-(pixel,row) scalar loads were ~3/4 of the epilogue's VMEM instructions.
    const int hi = lane >> 5, col = lane & 31;
    if (EPI == EPI_PLAIN) {
        const float isi = sh16_in_scale_inv(p);
        // f32 outputs of these convs use the "C4" layout [B][C/4][H][W][4]: a lane's 4 consecutive rows of one pixel are
        // one float4, lanes run along x -> 512 contiguous bytes per half-wave store (and per residual load).
        // Loop order: channel run (m, rq) outer, so that only one run's bias / scale float4 pair is live.
        const int C4n = (p.Mrows + 3) >> 2;
        const float slope = act_slope(p.act);
        const int rW = p.W >> p.res_up, rH = p.H >> p.res_up;
        int pb_[4], pix_[4];
        float amax = 0.f;              // max |out| (recorded in p.out_amax for consumers that split this f32 tensor: INC4)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            int tx, ty, tb;
            sh16_slot_px<TW, TH>(idx, rot, tx, ty, tb);
            const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
            pb_[n] = (b < p.B && y < p.H && x < p.W) ? b : -1;
            pix_[n] = D2S ? 4 * y * p.W + 2 * x : y * p.W + x;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int row0 = mtile64 * 64 + m * 32 + 8 * rq + 4 * hi;
                if (row0 >= p.Mrows) continue;                  // Mrows % 4 == 0
                if constexpr (D2S) {
                    const int Cr = p.Mrows >> 2, ph = row0 / Cr, co0 = row0 - ph * Cr;
                    const int poff = (ph >> 1) * 2 * p.W + (ph & 1);
                    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(isi, isi, isi, isi);
                    if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + co0);
                    if (p.wscale) {
                        const float4 t = *reinterpret_cast<const float4*>(p.wscale + row0);
                        s4 = make_float4(t.x * isi, t.y * isi, t.z * isi, t.w * isi);
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (pb_[n] < 0) continue;
                        float4 v;
                        v.x = acc[m][n][rq * 4 + 0] * s4.x + b4.x; v.y = acc[m][n][rq * 4 + 1] * s4.y + b4.y;
                        v.z = acc[m][n][rq * 4 + 2] * s4.z + b4.z; v.w = acc[m][n][rq * 4 + 3] * s4.w + b4.w;
                        v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
                        v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
                        reinterpret_cast<float4*>(p.out)[((long long)pb_[n] * (Cr >> 2) + (co0 >> 2)) * (4 * HW) + pix_[n] + poff] = v;
                    }
                    continue;
                }
                if (p.splitk > 1) {                              // split-K: raw partial sums to this slice's C4 slab
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (pb_[n] < 0) continue;
                        float4 v;
                        v.x = acc[m][n][rq * 4 + 0]; v.y = acc[m][n][rq * 4 + 1];
                        v.z = acc[m][n][rq * 4 + 2]; v.w = acc[m][n][rq * 4 + 3];
                        reinterpret_cast<float4*>(p.partial)[(((long long)ks * p.B + pb_[n]) * C4n + (row0 >> 2)) * HW + pix_[n]] = v;
                    }
                    continue;
                }
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(isi, isi, isi, isi);
                if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + row0);
                if (p.wscale) {
                    const float4 t = *reinterpret_cast<const float4*>(p.wscale + row0);
                    s4 = make_float4(t.x * isi, t.y * isi, t.z * isi, t.w * isi);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (pb_[n] < 0) continue;
                    const int b = pb_[n];
                    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.res) {
                        const int yy = pix_[n] / p.W, xx = pix_[n] - yy * p.W;
                        rv = reinterpret_cast<const float4*>(p.res)[((long long)b * C4n + (row0 >> 2)) * (rW * rH) +
                                                                    (yy >> p.res_up) * rW + (xx >> p.res_up)];
                    }
                    float4 v;
                    v.x = acc[m][n][rq * 4 + 0] * s4.x + b4.x; v.y = acc[m][n][rq * 4 + 1] * s4.y + b4.y;
                    v.z = acc[m][n][rq * 4 + 2] * s4.z + b4.z; v.w = acc[m][n][rq * 4 + 3] * s4.w + b4.w;
                    // act = max(v, slope*v): none / leaky / relu without a per-element switch
                    if (p.act > ACT_RELU) {                     // tanh / sigmoid (Zencoder head): rare, uniform branch
                        if (p.res_after_act) {
                            v.x = apply_act(v.x, p.act) + rv.x; v.y = apply_act(v.y, p.act) + rv.y;
                            v.z = apply_act(v.z, p.act) + rv.z; v.w = apply_act(v.w, p.act) + rv.w;
                        } else {
                            v.x = apply_act(v.x + rv.x, p.act); v.y = apply_act(v.y + rv.y, p.act);
                            v.z = apply_act(v.z + rv.z, p.act); v.w = apply_act(v.w + rv.w, p.act);
                        }
                    } else if (p.res_after_act) {
                        v.x = fmaxf(v.x, slope * v.x) + rv.x; v.y = fmaxf(v.y, slope * v.y) + rv.y;
                        v.z = fmaxf(v.z, slope * v.z) + rv.z; v.w = fmaxf(v.w, slope * v.w) + rv.w;
                    } else {
                        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        v.x = fmaxf(v.x, slope * v.x); v.y = fmaxf(v.y, slope * v.y);
                        v.z = fmaxf(v.z, slope * v.z); v.w = fmaxf(v.w, slope * v.w);
                    }
                    reinterpret_cast<float4*>(p.out)[((long long)b * C4n + (row0 >> 2)) * HW + pix_[n]] = v;
                    amax = fmaxf(fmaxf(amax, fabsf(v.x)), fabsf(v.y));
                    amax = fmaxf(fmaxf(amax, fabsf(v.z)), fabsf(v.w));
                }
            }
        if (p.out_amax && p.splitk <= 1) {
            amax = sh16_wave_max(amax);
            if (lane == 0) sh16_slot_max(p.out_amax, amax * SH16_ACT_SCALE);
        }
    } else {  // EPI_ACE -> SH16 output.  Loop order: channel-run (rq) outer so that only one run's parameters
              // (5 float4) are live; pixel coordinates / noise / packed 3x3 label neighbourhoods are kept per n.
        const int C = p.C;
        const int Go = (C + 7) >> 3;
        const int xW = p.W >> p.x_up, xH = p.H >> p.x_up;
        int pb_[4], py_[4], px_[4];
        float nzv[4];
        unsigned long long labs[4];          // 9 neighbour labels x 5 bits (19 = outside the image: the LUT's zero column)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            int tx, ty, tb;
            sh16_slot_px<TW, TH>(idx, rot, tx, ty, tb);
            const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
            const bool ok = b < p.B && y < p.H && x < p.W;
            pb_[n] = ok ? b : -1;
            py_[n] = y;
            px_[n] = x;
            nzv[n] = p.noise[ok ? (long long)b * p.noise_bstride + (long long)x * p.H + y : 0];
            unsigned long long lv = 0;
            if (p.lut && ok) {
                const uint8_t* lb = p.lab + (long long)b * HW;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                    const unsigned jv = lb[in ? yy * p.W + xx : 0];       // unconditional load, select after
                    lv |= (unsigned long long)((in && jv < 19u) ? jv : 19u) << (5 * t);   // >= 19: "no class" -> zero column
                }
            }
            labs[n] = lv;
        }
        // Addressing: wave-uniform 64-bit bases (sample b0) + 32-bit per-lane byte offsets.
        const float osc = p.out_scale != 0.f ? p.out_scale : 1.f, slope = act_slope(p.act);
        const float st = (p.wscale ? p.wscale[(mtile64 < p.mtiles ? mtile64 : 0) * 64] : 1.f) * (p.in_scale_inv != 0.f ? p.in_scale_inv : 1.f) * osc;
        const float extra = p.pass == 1 ? sh16_dyn_extra(*p.out_amax) : 1.f;
        float amax = 0.f;
        const int xHW = xW * xH;
        const char* xbase = reinterpret_cast<const char*>(p.x) + (long long)b0 * (C >> 2) * xHW * 16;
        char* obase = reinterpret_cast<char*>(p.out) + (long long)b0 * Go * 2 * HW * 16;
        const unsigned lrs = (unsigned)p.lut_rs * 4u, lns = (unsigned)p.lut_ns * 4u;      // LUT strides in bytes
        const char* lbase = reinterpret_cast<const char*>(p.lut) + (long long)b0 * p.lut_bs * lns;
        unsigned xo_[4], oo_[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned tb = pb_[n] < 0 ? 0u : (unsigned)(pb_[n] - b0);
            xo_[n] = (tb * (unsigned)(C >> 2) * xHW + (unsigned)(py_[n] >> p.x_up) * xW + (unsigned)(px_[n] >> p.x_up)) * 16u;
            oo_[n] = ((tb * Go * 2 + hi) * (unsigned)HW + (unsigned)py_[n] * p.W + (unsigned)px_[n]) * 16u;
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        auto body = [&](auto resc) {
            constexpr bool RESC = decltype(resc)::value;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int g = mtile64 * 4 + rq;                  // output channel group (8 channels)
                const int c0 = g * 8 + 4 * hi;                   // this lane's 4 consecutive channels (C % 4 == 0)
                if (g >= Go) continue;
                const bool cok = c0 < C;
                const int cc = cok ? c0 : 0;
                float4 pg = *reinterpret_cast<const float4*>(p.bias_g + cc);
                float4 pb = *reinterpret_cast<const float4*>(p.bias_b + cc);
                const float4 pa = *reinterpret_cast<const float4*>(p.bn_a + cc);
                const float4 pd = *reinterpret_cast<const float4*>(p.bn_d + cc);
                const float4 pn = *reinterpret_cast<const float4*>(p.nv + cc);
                // the output scale is folded into the constants (the style LUT is stored pre-multiplied by it)
                pg = make_float4((pg.x + 1.f) * osc, (pg.y + 1.f) * osc, (pg.z + 1.f) * osc, (pg.w + 1.f) * osc);
                pb = make_float4(pb.x * osc, pb.y * osc, pb.z * osc, pb.w * osc);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (pb_[n] < 0) continue;
                    const unsigned tb = (unsigned)(pb_[n] - b0);
                    float4 sg = z4, sb = z4;
                    if (p.lut) {
                        // style term: sum over the 9 taps of LUT[(sample, label at the tap)][tap]; taps outside the image
                        // carry label 19, the all-zero column.  32-bit offsets from a wave-uniform base.
                        const unsigned cb = tb * (unsigned)p.lut_bs * lns + (unsigned)cc * lrs;
                        const unsigned lo30 = (unsigned)labs[n], hi15 = (unsigned)(labs[n] >> 30);
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const unsigned j = t < 6 ? (lo30 >> (5 * t)) & 31u : (hi15 >> (5 * (t - 6))) & 31u;
                            const unsigned o1 = cb + j * lns + (unsigned)(t * 2 * C) * lrs;
                            const float4 g4 = *reinterpret_cast<const float4*>(lbase + o1);
                            const float4 b4 = *reinterpret_cast<const float4*>(lbase + (o1 + (unsigned)C * lrs));
                            sg.x += g4.x; sg.y += g4.y; sg.z += g4.z; sg.w += g4.w;
                            sb.x += b4.x; sb.y += b4.y; sb.z += b4.z; sb.w += b4.w;
                        }
                    }
                    // x in the C4 layout [B][C/4][h][w][4]: this lane's 4 channels of the pixel are one float4
                    const float4 x4 = *reinterpret_cast<const float4*>(xbase + (xo_[n] + (unsigned)(cc >> 2) * xHW * 16u));
                    const float4 bg = make_float4(pg.x + sg.x, pg.y + sg.y, pg.z + sg.z, pg.w + sg.w);
                    const float4 bb = make_float4(pb.x + sb.x, pb.y + sb.y, pb.z + sb.z, pb.w + sb.w);
                    uint4 w = ace_quad<RESC, BF>(acc[0][n][rq * 4 + 0], acc[0][n][rq * 4 + 1], acc[0][n][rq * 4 + 2], acc[0][n][rq * 4 + 3],
                                             acc[1][n][rq * 4 + 0], acc[1][n][rq * 4 + 1], acc[1][n][rq * 4 + 2], acc[1][n][rq * 4 + 3],
                                             st, bg, bb, pa, pd, pn, x4, nzv[n], slope, extra, amax);
                    if (!cok) w = make_uint4(0, 0, 0, 0);            // padding channels of the last group hold zeros
                    *reinterpret_cast<uint4*>(obase + (oo_[n] + (unsigned)g * 2u * (unsigned)HW * 16u)) = sh16_pair_swap(w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (p.pass == 1) {
            body(std::true_type{});
        } else {
            body(std::false_type{});
            if (p.out_amax) {
                amax = sh16_wave_max(amax);
                if (lane == 0) sh16_slot_max(p.out_amax, amax);
            }
        }
    }
}

// in : SH16 [B][Cin/8][2][H][W][8]   (Cin % 16 == 0; padding channels hold zeros)
// EPI_PLAIN -> out f32 NCHW [B][Mrows][H][W] (bias / residual / act as conv_mfma)
// EPI_ACE   -> out SH16 [B][ceil(C/8)][2][H][W][8]  (the fused ACE epilogue of conv_mfma.h, re-split for the next conv)
// INC4: `in` is an f32 tensor in the C4 layout [B][Cin/4][H][W][4] (what the EPI_PLAIN epilogue writes) and is split into
// f16 (hi, lo) units while it is staged: two float4 loads per (8-channel group, pixel) instead of two 16-byte SH16 units,
// i.e. the same bytes and no separate conversion pass.  Scale: SH16_ACT_SCALE x the dynamic factor of the producer's
// recorded maximum (p.in_amax; complete, because the producer is an earlier kernel), so the units always sit in the f16
// window -- no second pass.  Used by the BiSeNet trunk (conv -> BN -> ReLU chains with residuals, all in C4).
// S2D (stride-2 convs, KS = 2 or 1): the conv runs at the OUTPUT resolution over the space-to-depth view of the input.
// With the k x k, stride 2, pad 1 kernel indexed by ky = 2 dy + py (dy in {0,1}, py in {0,1}), output pixel y reads physical
// row 2 (y + dy) + py - 1: a 2x2-tap stride-1 conv over 4 "phase" copies of the input channels (phase = (py, px); k = 4:
// all 16 (tap, phase) pairs carry a weight, k = 3: 9 of 16).  The view is address arithmetic in the staging: chunk ->
// (phase, 16 real channels), unit -> physical pixel 2 (y, x) + phase - 1, zero outside the image.  KS = 1: the 1x1 stride-2
// shortcut convs (phase (1,1) only = physical pixel 2 (y, x)).  p.Cin = phases x p.s2d_cr, p.H / p.W = output size.
template <int KS, int TW, int TH, int TB, int EPI, int TERMS = 3, bool FUSE = false, bool INC4 = false, bool S2D = false,
          bool D2S = false>
__global__ __launch_bounds__(256, 2) void conv_sh16_kernel(const ConvParams p) {
    static_assert(!D2S || (KS == 2 && EPI == EPI_PLAIN && !FUSE && !S2D), "D2S: the 2x2-tap form of ConvTranspose2d(k3, s2)");
    static_assert(!(FUSE && INC4), "the fused second operand is an SH16 tensor");
    static_assert(!S2D || (!FUSE && EPI == EPI_PLAIN && (KS == 2 || KS == 1)), "S2D: plain 2x2-tap / 1x1 convs only");
    static_assert(S2D || D2S || KS != 2, "2x2 taps exist for the space-to-depth / depth-to-space forms only");
    using Cfg = ShCfg<KS, TW, TH, TB>;
    constexpr int PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE, UNITS = Cfg::UNITS, NLOAD = Cfg::NLOAD, HALO = Cfg::HALO;
    constexpr int NT = KS * KS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
    if constexpr (EPI == EPI_ACE) {
        sh16_mode_on();                                      // saturating f32 -> f16 conversions in the epilogue
        if (p.pass == 1 && sh16_dyn_extra(*p.out_amax) == 1.f) return;   // second pass: nothing to repair (the normal case)
    }
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile64 = L % p.mtiles;
    int nt = L / p.mtiles;
    const bool split = EPI == EPI_PLAIN && p.splitk > 1;
    const int ks = split ? nt % p.splitk : 0;            // K slice of this block (split-K, low-resolution layers)
    if (split) nt /= p.splitk;
    const int c_lo = split ? ks * p.cps : 0;
    const int c_hi = split ? (c_lo + p.cps < p.nchunks ? c_lo + p.cps : p.nchunks) : p.nchunks;
    const int txi = nt % p.tiles_x; nt /= p.tiles_x;
    const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH, b0 = nt * TB;
    const int HW = p.H * p.W;
    const int G = p.Cin >> 3;                            // input channel groups

    constexpr int ROT = TW == 16 ? (16 - PW % 16) % 16 : 0;     // bank-conflict-free lane -> pixel mapping (sh16_slot_px)
    int ub[4];   // per-lane LDS unit offset of the window origin (hi plane of this lane's k-half) per N-subtile
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        int tx, ty, tb;
        sh16_slot_px<TW, TH>(idx, ROT, tx, ty, tb);
        ub[n] = (lane >> 5) * 2 * PLANE + tb * (PH * PW) + ty * PW + tx;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const uint4* gin = reinterpret_cast<const uint4*>(p.in);
    // physical plane size of the input: ConvTranspose2d(k3,s2,p1,op1) runs as a conv over the zero-inserted x2 view
    // (IN_UP2_NEAREST: the shape decoder's nearest x2 up-sampling, shape_branch/model.py:128, as address arithmetic)
    const int HWi = S2D ? 4 * HW : (p.in_mode != IN_DIRECT ? (p.H >> 1) * (p.W >> 1) : HW);
    // S2D: validity of a patch position per phase (bit ph set = physical pixel 2 (y, x) + (ph >> 1, ph & 1) - 1 lies in the
    // image); soff then holds the (possibly negative) offset of phase (0, 0) and is only used under a set bit
    [[maybe_unused]] int sval[NLOAD];
    [[maybe_unused]] const int Wp = 2 * p.W, Hp = 2 * p.H, ncr = S2D ? p.s2d_cr >> 4 : 1;
    auto s2d_mask = [&](int y, int x, int b, int py, int px) {
        int m = 0;
        if (b < p.B && (KS == 1 || (py >= 1 && px >= 1)))           // KS == 2: patch row / column 0 is never addressed
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int yy = 2 * y - 1 + (ph >> 1), xx = 2 * x - 1 + (ph & 1);
                if ((unsigned)yy < (unsigned)Hp && (unsigned)xx < (unsigned)Wp) m |= 1 << ph;
            }
        return m;
    };
    // per-thread source offsets of its NLOAD patch units (chunk-invariant part), -1 = outside the image -> zeros.
    // Hoisted out of the chunk loop: the decode (3 div/mod per unit) was ~half of the kernel's VALU issue.
    int soff[NLOAD];
    constexpr int NPAIR = (2 * PLANE + 255) / 256;             // INC4: (group, pixel) pairs per thread
    float c4_scale = 1.f;
    if constexpr (INC4) {
        sh16_mode_on();
        c4_scale = (p.in_scale_inv != 0.f ? 1.f / p.in_scale_inv : 1.f) * (p.in_amax ? sh16_dyn_extra(*p.in_amax) : 1.f);
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
            const int q = tid + i * 256;
            soff[i] = -1;
            if constexpr (S2D) sval[i] = 0;
            if (q < 2 * PLANE) {
                const int g = q / PLANE, rem = q % PLANE;
                const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
                int y = y0 + py - HALO, x = x0 + px - HALO;
                const int b = b0 + tb;
                if constexpr (S2D) {
                    sval[i] = s2d_mask(y, x, b, py, px);
                    soff[i] = (b * (p.s2d_cr >> 2) + g * 2) * HWi + (2 * y - 1) * Wp + (2 * x - 1);
                    continue;
                }
                if (p.pad_mode == PAD_REFLECT) {
                    y = y < 0 ? -y : (y >= p.H ? 2 * (p.H - 1) - y : y);
                    x = x < 0 ? -x : (x >= p.W ? 2 * (p.W - 1) - x : x);
                }
                if (b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                    // float4 index of the group's first C4 plane (4 planes per 16-channel chunk)
                    if (p.in_mode == IN_UP2_NEAREST) {
                        soff[i] = (b * (p.Cin >> 2) + g * 2) * HWi + (y >> 1) * (p.W >> 1) + (x >> 1);
                    } else if (p.in_mode == IN_UP2_ZEROINS) {
                        if (!((y | x) & 1)) soff[i] = (b * (p.Cin >> 2) + g * 2) * HWi + (y >> 1) * (p.W >> 1) + (x >> 1);
                    } else {
                        soff[i] = (b * (p.Cin >> 2) + g * 2) * HW + y * p.W + x;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < (INC4 ? 0 : NLOAD); ++i) {
        const int u = tid + i * 256;
        soff[i] = -1;
        if constexpr (S2D) sval[i] = 0;
        if (u < UNITS) {
            const int gh = u / PLANE, rem = u % PLANE;          // gh = group*2 + hl
            const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
            int y = y0 + py - HALO, x = x0 + px - HALO;
            const int b = b0 + tb;
            if constexpr (S2D) {
                sval[i] = (TERMS != 3 && (gh & 1)) ? 0 : s2d_mask(y, x, b, py, px);
                soff[i] = ((b * (p.s2d_cr >> 3) + (gh >> 1)) * 2 + (gh & 1)) * HWi + (2 * y - 1) * Wp + (2 * x - 1);
                continue;
            }
            if (p.pad_mode == PAD_REFLECT) {      // nn.ReflectionPad2d (Zencoder, architecture.py:174)
                y = y < 0 ? -y : (y >= p.H ? 2 * (p.H - 1) - y : y);
                x = x < 0 ? -x : (x >= p.W ? 2 * (p.W - 1) - x : x);
            }
            if (b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
                if (p.in_mode == IN_UP2_NEAREST) {
                    soff[i] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HWi + (y >> 1) * (p.W >> 1) + (x >> 1);
                } else if (p.in_mode == IN_UP2_ZEROINS) {   // logical input = physical input with zeros between the samples
                    if (!((y | x) & 1)) soff[i] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HWi + (y >> 1) * (p.W >> 1) + (x >> 1);
                } else {
                    soff[i] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
                }
            }
            if (TERMS != 3 && (gh & 1)) soff[i] = -1;           // single-term f16 path never reads the lo planes
        }
    }
    // FUSE: the two inputs share the accumulators (sh16_in_scale_inv).  mul1 = mul2 = 1 unless a second pass changed one of
    // the tensors' scales; then the units are rescaled while they are staged (through f32: the factor may be far outside
    // the f16 range; wave-uniform branch, not taken in the normal case).
    float mul1 = 1.f, mul2 = 1.f;
    if constexpr (FUSE) {
        sh16_mode_on();
        (void)sh16_in_scale_inv(p, &mul1, &mul2);
    }
    auto rescale = [&](uint4 (&stg)[NLOAD], float f) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            half8 h = __builtin_bit_cast(half8, stg[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)((float)h[e] * f);
            stg[i] = __builtin_bit_cast(uint4, h);
        }
    };
    auto stage = [&](int chunk, int buf) {
        if constexpr (INC4) {
            float4 sa[NPAIR], sc[NPAIR];
            int phase = 0, rc = chunk;
            if constexpr (S2D) {
                phase = chunk / ncr;
                rc = chunk - phase * ncr;
                phase += p.s2d_phase0;
            }
            const float4* src = reinterpret_cast<const float4*>(p.in) + (long long)rc * 4 * HWi + (S2D ? (phase >> 1) * Wp + (phase & 1) : 0);
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                bool ok = soff[i] >= 0;
                if constexpr (S2D) ok = (sval[i] >> phase) & 1;
                sa[i] = ok ? src[soff[i]] : make_float4(0.f, 0.f, 0.f, 0.f);
                sc[i] = ok ? src[soff[i] + HWi] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            uint4* dst = smem_u + buf * UNITS;
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                const int q = tid + i * 256;
                if (q < 2 * PLANE) {
                    const int g = q / PLANE, rem = q - g * PLANE;
                    const float v[8] = {sa[i].x, sa[i].y, sa[i].z, sa[i].w, sc[i].x, sc[i].y, sc[i].z, sc[i].w};
                    half8 h, l;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        _Float16 he, le;
                        sh16_split(v[e], c4_scale, he, le);
                        h[e] = he;
                        l[e] = le;
                    }
                    dst[(g * 2) * PLANE + rem] = __builtin_bit_cast(uint4, h);
                    dst[(g * 2 + 1) * PLANE + rem] = __builtin_bit_cast(uint4, l);
                }
            }
            return;
        }
        // No scheduling fence in here on purpose: the compiler issues these loads early and sinks the LDS writes
        // below the chunk's MFMAs as far as registers allow, which is what overlaps staging with compute.
        uint4 stg[NLOAD];
        if constexpr (S2D) {
            const int ph0 = chunk / ncr, phase = ph0 + p.s2d_phase0;
            const uint4* src = gin + (long long)(chunk - ph0 * ncr) * 4 * HWi + (phase >> 1) * Wp + (phase & 1);
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) stg[i] = ((sval[i] >> phase) & 1) ? src[soff[i]] : make_uint4(0, 0, 0, 0);
        } else {
            const uint4* src = gin + (long long)chunk * 4 * HWi;      // 2 groups x (hi, lo) planes per chunk
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) stg[i] = soff[i] >= 0 ? src[soff[i]] : make_uint4(0, 0, 0, 0);
        }
        if constexpr (FUSE) {
            if (mul1 != 1.f) rescale(stg, mul1);
        }
        uint4* dst = smem_u + buf * UNITS;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < UNITS) dst[u] = stg[i];
        }
    };

    // A fragments: [mtile][chunk][tap][msub][hl][lane] units of 16 B
    const uint4* Ap = reinterpret_cast<const uint4*>(p.wpk) + ((long long)mtile64 * p.nchunks) * (NT * 4 * 64) + lane;
    // FUSE: second operand (see the loop after the main loop); its first chunk is staged during the last 3x3 chunk
    const int nch2 = FUSE ? (p.Cin2 + 15) >> 4 : 0;
    const uint4* gin2 = reinterpret_cast<const uint4*>(p.in2) + (FUSE ? (long long)b0 * ((p.Cin2 >> 3) - G) * 2 * HW : 0);
    const uint4* Ap2 = reinterpret_cast<const uint4*>(p.wpk2) + ((long long)mtile64 * nch2) * (4 * 64) + lane;
    auto stage2 = [&](int chunk, int buf) {
        uint4 stg[NLOAD];
        const uint4* src = gin2 + (long long)chunk * 4 * HW;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) stg[i] = soff[i] >= 0 ? src[soff[i]] : make_uint4(0, 0, 0, 0);
        if (mul2 != 1.f) rescale(stg, mul2);
        uint4* dst = smem_u + buf * UNITS;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < UNITS) dst[u] = stg[i];
        }
    };

    if (!CH_ABL(p.dbg & 8)) stage(c_lo, 0);
    __syncthreads();

    for (int ch = c_lo; ch < c_hi; ++ch) {
        if (ch + 1 < c_hi && !CH_ABL(p.dbg & 1)) stage(ch + 1, (ch + 1 - c_lo) & 1);
        if constexpr (FUSE) {
            if (ch + 1 == c_hi) stage2(0, (ch + 1 - c_lo) & 1);
        }
        const uint4* sb = smem_u + ((ch - c_lo) & 1) * UNITS;
        if (CH_ABL(p.dbg & 2)) { __syncthreads(); continue; }
        const uint4* Ac = Ap + (long long)ch * (NT * 4 * 64);
        uint4 a_cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = Ac[q * 64];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4 a_nxt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a_nxt[q] = a_cur[q];
            if (t + 1 < NT) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a_nxt[q] = Ac[((t + 1) * 4 + q) * 64];
            }
            const int koff = KS == 2 ? (1 + t / 2) * PW + (1 + t % 2) : (t / KS) * PW + (t % KS);   // KS 2: window (y, x)..(y+1, x+1)
            asm volatile("" ::: "memory");          // no cross-tap CSE of LDS reads
            uint4 bh[4], bl[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bh[n] = sb[ub[n] + koff];
                bl[n] = sb[ub[n] + koff + PLANE];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint4 ah = a_cur[m * 2 + 0], al = a_cur[m * 2 + 1];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (TERMS == 3) {
                        acc[m][n] = mfma16<TERMS>(al, bh[n], acc[m][n]);
                        acc[m][n] = mfma16<TERMS>(ah, bl[n], acc[m][n]);
                    }
                    acc[m][n] = mfma16<TERMS>(ah, bh[n], acc[m][n]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
        }
        __syncthreads();
    }

    if constexpr (FUSE) {
        // Fused 1x1 operand (ResBlock shortcut conv_s folded into conv_1: out = W1 (3x3) h1 + Ws (1x1) hs + b): the same
        // pixel patch of the SECOND input, 16 channels per chunk, centre tap only, into the same accumulators -- saves the
        // shortcut's own kernel, its C4 write and the residual read.  TB == 1 (one sample per block), no split-K.
        for (int c2 = 0; c2 < nch2; ++c2) {
            const int v = p.nchunks + c2;                      // virtual chunk index: LDS stage parity continues
            if (c2 + 1 < nch2) stage2(c2 + 1, (v + 1) & 1);
            const uint4* sb = smem_u + (v & 1) * UNITS;
            const uint4* Ac2 = Ap2 + (long long)c2 * (4 * 64);
            uint4 a2[4], bh[4], bl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) a2[q] = Ac2[q * 64];
            constexpr int kc = HALO * PW + HALO;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bh[n] = sb[ub[n] + kc];
                bl[n] = sb[ub[n] + kc + PLANE];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint4 ah = a2[m * 2 + 0], al = a2[m * 2 + 1];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (TERMS == 3) {
                        acc[m][n] = mfma16<TERMS>(al, bh[n], acc[m][n]);
                        acc[m][n] = mfma16<TERMS>(ah, bl[n], acc[m][n]);
                    }
                    acc[m][n] = mfma16<TERMS>(ah, bh[n], acc[m][n]);
                }
            }
            __syncthreads();
        }
    }

    if (CH_ABL(p.dbg & 4)) return;
    sh16_epilogue<TW, TH, TB, EPI, TERMS == 2, D2S>(p, acc, mtile64, wn, lane, x0, y0, b0, ks, ROT);
}

// -----------------------------------------------------------------------------------------------------------------
// Wave-specialised persistent variant (large layers).  One 512-thread block per CU: waves 0-3 are CONSUMERS (LDS
// fragment reads + MFMA + epilogue), waves 4-7 are LOADERS: input patches HBM/L2 -> VGPR -> LDS three chunks ahead (two
// register sets), A fragments L2 -> LDS by DMA (global_load_lds), both into a 2-stage ring.  A wave's vector-memory
// results return in order, so a wave that mixes multi-microsecond patch loads with its A-fragment loads stalls its MFMA
// stream on every chunk; here the consumers issue no vector-memory load at all inside the main loop.  Blocks are
// persistent (grid = #CUs) and walk the tile list with a static stride, so the loaders already stage the next tile's
// first chunks while the consumers run the epilogue (no exposed prologue).
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// CP (EPI_ACE, 32x16 tiles): pixel-level compaction of the exact SPADE-interior reduction (ace_sparse.h).  The loaders still
// stage the whole patch and the A fragments of every listed tile, but the consumers run their MFMAs and the ACE epilogue over the
// tile's BOUNDARY pixels only: the per-tile list (uint16 in-tile offsets, raster order) and count reach LDS a tile ahead
// (double-buffered, staged by the loaders), a consumer wave takes the 32-pixel sub-tiles wn, wn + 4, ... and reads its B fragments
// at the patch offsets of ITS pixels.  Operand delivery (76 KB per chunk at ~12 B/cycle/CU) then bounds a sparse tile's k-loop,
// not the MFMAs; the epilogue shrinks with the pixel count.
template <int KS, int TW, int TH, int TB, int EPI, int TERMS = 3, int CP = 0>      // CP: 0 / 1 compacting / 2 compacting, pair entries
__global__ __launch_bounds__(512, 2) void conv_sh16_ws_kernel(const ConvParams p) {
    static_assert(!CP || (EPI == EPI_ACE && KS == 3 && TW == 32 && TH == 16 && TB == 1), "CP: the ACE kernel's 32x16 tiles only");
    using Cfg = ShCfg<KS, TW, TH, TB>;
    constexpr int PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE, UNITS = Cfg::UNITS, HALO = Cfg::HALO;
    constexpr int NT = KS * KS;
    constexpr int NLD = (UNITS + 255) / 256;                 // units per loader thread per chunk
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
    // LDS map (16-byte units): 2 stages of [input patch UNITS | A fragments AUNITS], then (ACE) the epilogue's small
    // operands [par: 8 runs x 5 float4][nz: 512 f32][lab: TB*(TH+2)*(TW+2) bytes].  The A fragments of a chunk are one
    // contiguous, already lane-ordered block in global memory: the loaders move it with LDS-DMA (global_load_lds, 16 B
    // per lane, no VGPR round trip), so the consumers' MFMA stream never waits on an L2 round trip.
    constexpr int AUNITS = NT * 4 * 64, STAGE = UNITS + AUNITS;
    constexpr int NPAR = 5;                                  // float4 per channel run: s(1+bias_g), s*bias_b, bn_a, bn_d, nv
    constexpr bool PAIR = CP >= 2;                           // CP 2 / 3: entries of two / four row tiles (MG), direct A stream
    constexpr int MG = CP == 3 ? 4 : (CP == 2 ? 2 : 1);
    constexpr int NRUN = 8 * MG;                             // the epilogue parameters of MG 64-row tiles
    constexpr int PAR0 = 2 * STAGE, NZ0 = PAR0 + NRUN * NPAR, LAB0 = NZ0 + 128;
    [[maybe_unused]] constexpr int LIST0 = LAB0 + (TB * (TH + 2) * (TW + 2) + 15) / 16, META0 = LIST0 + 128;   // CP: 2 x 1 KB lists, 2 x count
    constexpr int NDA = AUNITS / 256;                        // A DMA instructions per loader thread per chunk
    constexpr int LW = TW + 2, LH = TH + 2;
    constexpr bool pre = EPI == EPI_ACE;    /* host guarantees nchunks >= 3 */       // epilogue operands prefetched through LDS

    if constexpr (EPI == EPI_ACE) {
        sh16_mode_on();                                      // saturating f32 -> f16 conversions in the epilogue
        if (p.pass == 1 && sh16_dyn_extra(*p.out_amax) == 1.f) return;   // second pass: nothing to repair (the normal case)
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int wn = wave & 3, ltid = tid & 255;
    const int HW = p.H * p.W;
    const int G = p.Cin >> 3;
    // Tile-skip mode of the exact SPADE-interior reduction (ace_sparse.h; EPI_ACE only): p.sp_work lists the (spatial tile, row
    // tile) pairs whose spatial tile holds at least one boundary pixel -- tiles made of interior pixels only are served by
    // ace_interior_sh16 and never reach this kernel.  Same static stride, over the list instead of the full tile grid.
    const bool sp = EPI == EPI_ACE && p.sp_work != nullptr;
    const int ntiles = sp ? p.sp_total[0] : p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    const int first = xcd_remap(blockIdx.x, gridDim.x);      // logical slot of this block inside one round of tiles
    const int my_tiles = first < ntiles ? (ntiles - 1 - first) / (int)gridDim.x + 1 : 0;
    const int Q = my_tiles * p.nchunks;                       // chunks this block will process
    const uint4* gin = reinterpret_cast<const uint4*>(p.in);

    auto tile_coords = [&](int k, int& mtile64, int& x0, int& y0, int& b0) {
        const int L = first + k * (int)gridDim.x;
        int nt;
        if (sp) {
            const unsigned wk = p.sp_work[L];
            mtile64 = (int)(wk >> 20);
            if (PAIR) mtile64 *= MG;                              // pair / quad entry: first of its row tiles
            nt = (int)(wk & 0xFFFFFu);
        } else {
            mtile64 = L % p.mtiles;
            nt = L / p.mtiles;
        }
        const int txi = nt % p.tiles_x; nt /= p.tiles_x;
        const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
        x0 = txi * TW; y0 = tyi * TH; b0 = nt * TB;
    };

    // CP == 2, pair entries (ace_worklist mode 3: spatial tiles with at most four 32-pixel sub-tiles of boundary pixels): one
    // iteration serves TWO row tiles of the spatial tile.  Waves 0-1 own row tile 2m, waves 2-3 row tile 2m + 1, each wave at
    // most two sub-tiles, and every wave streams the A fragments of ITS row tile from L2 straight into registers (kloop_direct)
    // -- nothing goes through the loaders' A DMA, whose issue -> land -> barrier chain (~4.5 k cycles per chunk) bounded these
    // tiles at one sub-tile (1.7 k cycles of MFMA) per wave; the staged patch now feeds twice the rows.
    if (loader) {
        if constexpr (CP) {
            // ------------------------------------------------------------ loaders of the compacting kernel: split roles
            // A wave's vector-memory results return in order.  In the shared loader below every wave issues both the chunk's A
            // DMA and its patch prefetch and must see the DMA land inside the chunk's own iteration -- which also drains the
            // patch loads issued one iteration earlier, i.e. gives them ONE chunk time to return.  On compacted tiles a chunk
            // is 4x shorter than an HBM round trip under load, and the k-loop ran at that latency (39 k cycles for 13.8 k of
            // MFMAs).  Here waves 4-5 only move A fragments (+ the small epilogue operands and the pixel lists) and waves 6-7
            // only prefetch patches three chunks ahead, each with its own vmcnt.
            const int lw = wave - 4, ht = ltid & 127;                     // role-local thread id 0..127
            if (lw < 2) {
                const uint4* gA = reinterpret_cast<const uint4*>(p.wpk);
                int dma_k = -1, dma_mt = 0;
                auto dmaA = [&](int q) {
                    const int kq = q / p.nchunks;
                    if (kq != dma_k) {
                        int x0, y0, b0;
                        tile_coords(kq, dma_mt, x0, y0, b0);
                        dma_k = kq;
                    }
                    if constexpr (PAIR) return;                         // the consumers read the A fragments themselves
                    const uint4* src = gA + ((long long)dma_mt * p.nchunks + q % p.nchunks) * AUNITS + ht;
                    uint4* dst = smem_u + (q & 1) * STAGE + UNITS + lw * 64;
#pragma unroll
                    for (int i = 0; i < AUNITS / 128; ++i)
                        __builtin_amdgcn_global_load_lds((glb_void*)(src + i * 128), (lds_void*)(dst + i * 128), 16, 0, 0);
                };
                constexpr int NPI = (NRUN * NPAR + 127) / 128;
                float4 parr[NPI];
#pragma unroll
                for (int pi = 0; pi < NPI; ++pi) parr[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
                float nzr[4] = {0.f, 0.f, 0.f, 0.f};
                uint8_t labr[5] = {255, 255, 255, 255, 255};
                unsigned listr[2] = {0u, 0u};
                int cntr = 0;
                auto small_load = [&](int k) {                            // epilogue operands of tile k, pixel list of tile k + 1
                    int mt, x0, y0, b0;
                    tile_coords(k, mt, x0, y0, b0);
                    const int C = p.C;
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi) {                       // (runs 8 and up: the further row tiles of a pair / quad entry)
                        const int e = ht + pi * 128;
                        if (e >= NRUN * NPAR) break;
                        const int run = e / NPAR, which = e % NPAR;
                        const int c0 = (mt * 8 + run) * 4;
                        const float* src = which == 0 ? p.bias_g : (which == 1 ? p.bias_b : (which == 2 ? p.bn_a : (which == 3 ? p.bn_d : p.nv)));
                        float4 pv = *reinterpret_cast<const float4*>(src + (c0 < C ? c0 : 0));
                        const float osc = p.out_scale != 0.f ? p.out_scale : 1.f;
                        if (which == 0) pv = make_float4((pv.x + 1.f) * osc, (pv.y + 1.f) * osc, (pv.z + 1.f) * osc, (pv.w + 1.f) * osc);
                        if (which == 1) pv = make_float4(pv.x * osc, pv.y * osc, pv.z * osc, pv.w * osc);
                        parr[pi] = pv;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = ht + i * 128;
                        const int y = y0 + idx / TW, x = x0 + idx % TW;
                        const bool ok = b0 < p.B && y < p.H && x < p.W;
                        nzr[i] = p.noise[ok ? (long long)b0 * p.noise_bstride + (long long)x * p.H + y : 0];
                    }
                    if (p.lut) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            const int e = ht + i * 128;
                            labr[i] = 255;
                            if (e < LH * LW) {
                                const int y = y0 - 1 + e / LW, x = x0 - 1 + e % LW;
                                const bool in = b0 < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                                const uint8_t v = p.lab[in ? (long long)b0 * HW + y * p.W + x : 0];
                                labr[i] = in ? v : (uint8_t)255;
                            }
                        }
                    }
                };
                auto small_store = [&]() {
#pragma unroll
                    for (int pi = 0; pi < NPI; ++pi)
                        if (ht + pi * 128 < NRUN * NPAR) reinterpret_cast<float4*>(smem_u + PAR0)[ht + pi * 128] = parr[pi];
#pragma unroll
                    for (int i = 0; i < 4; ++i) reinterpret_cast<float*>(smem_u + NZ0)[ht + i * 128] = nzr[i];
                    if (p.lut) {
#pragma unroll
                        for (int i = 0; i < 5; ++i) {
                            const int e = ht + i * 128;
                            if (e < LH * LW) reinterpret_cast<uint8_t*>(smem_u + LAB0)[e] = labr[i];
                        }
                    }
                };
                auto list_load = [&](int k) {
                    const unsigned tile = p.sp_work[first + k * (int)gridDim.x] & 0xFFFFFu;
                    const unsigned* src = reinterpret_cast<const unsigned*>(p.sp_list + (long long)tile * (TW * TH));
                    listr[0] = src[ht];
                    listr[1] = src[ht + 128];
                    if (ht == 0) cntr = p.sp_cnt[tile];
                };
                auto list_store = [&](int k) {
                    unsigned* dst = reinterpret_cast<unsigned*>(smem_u + LIST0 + (k & 1) * 64);
                    dst[ht] = listr[0];
                    dst[ht + 128] = listr[1];
                    if (ht == 0) reinterpret_cast<int*>(smem_u + META0)[k & 1] = cntr;
                };
                if (Q > 0) {
                    dmaA(0);
                    list_load(0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (Q > 0) list_store(0);
                __syncthreads();                                  // stage 0 ready
                for (int q = 0; q < Q; ++q) {
                    const int k = q / p.nchunks, ch = q % p.nchunks;
                    if (q + 1 < Q) dmaA(q + 1);
                    if (ch == p.nchunks - 1) {                    // consumers read these after this iteration's barrier
                        small_store();
                        if (k + 1 < my_tiles) list_store(k + 1);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the A fragments of chunk q + 1 have landed
                    if (ch == p.nchunks - 2) {
                        small_load(k);
                        if (k + 1 < my_tiles) list_load(k + 1);
                    }
                    __syncthreads();
                }
            } else {
                constexpr int NL2 = (UNITS + 127) / 128;
                int soff[NL2];
                int cur_tile = -1;
                auto set_tile = [&](int k) {
                    int mt, x0, y0, b0;
                    tile_coords(k, mt, x0, y0, b0);
#pragma unroll
                    for (int i = 0; i < NL2; ++i) {
                        const int u = ht + i * 128;
                        soff[i] = -1;
                        if (u < UNITS) {
                            const int gh = u / PLANE, rem = u % PLANE;
                            const int py = rem / PW, px = rem % PW;
                            const int y = y0 + py - HALO, x = x0 + px - HALO;
                            if (b0 < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                                soff[i] = ((b0 * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
                        }
                    }
                    cur_tile = k;
                };
                uint4 stgA[NL2], stgB[NL2];
                auto load_chunk = [&](int q, uint4 (&stg)[NL2]) {
                    const int k = q / p.nchunks, ch = q % p.nchunks;
                    if (k != cur_tile) set_tile(k);
                    const uint4* src = gin + (long long)ch * 4 * HW;
#pragma unroll
                    for (int i = 0; i < NL2; ++i) {
                        const uint4 v = src[soff[i] >= 0 ? soff[i] : 0];
                        stg[i] = soff[i] >= 0 ? v : make_uint4(0, 0, 0, 0);
                    }
                };
                auto store_chunk = [&](int stage, const uint4 (&stg)[NL2]) {
                    uint4* dst = smem_u + stage * STAGE;
#pragma unroll
                    for (int i = 0; i < NL2; ++i) {
                        const int u = ht + i * 128;
                        if (u < UNITS) dst[u] = stg[i];
                    }
                };
                if (Q > 0) {
                    load_chunk(0, stgA);
                    if (Q > 1) load_chunk(1, stgB);
                    store_chunk(0, stgA);
                    if (Q > 2) load_chunk(2, stgA);
                }
                __syncthreads();                                  // stage 0 ready
                auto iter = [&](int q, uint4 (&stg)[NL2]) {
                    if (q + 1 < Q) store_chunk((q + 1) & 1, stg);  // requested two iterations ago
                    if (q + 3 < Q) load_chunk(q + 3, stg);
                    __syncthreads();
                };
                for (int q = 0; q < Q; q += 2) {
                    iter(q, stgB);
                    if (q + 1 < Q) iter(q + 1, stgA);
                }
            }
            return;
        }
        // ---------------------------------------------------------------- loaders
        int soff[NLD];
        int cur_tile = -1;
        auto set_tile = [&](int k) {
            int mt, x0, y0, b0;
            tile_coords(k, mt, x0, y0, b0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int u = ltid + i * 256;
                soff[i] = -1;
                if (u < UNITS) {
                    const int gh = u / PLANE, rem = u % PLANE;
                    const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
                    int y = y0 + py - HALO, x = x0 + px - HALO;
                    const int b = b0 + tb;
                    if (p.pad_mode == PAD_REFLECT) {
                        y = y < 0 ? -y : (y >= p.H ? 2 * (p.H - 1) - y : y);
                        x = x < 0 ? -x : (x >= p.W ? 2 * (p.W - 1) - x : x);
                    }
                    if (b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                        soff[i] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
                    if (TERMS != 3 && (gh & 1)) soff[i] = -1;           // single-term f16 path never reads the lo planes
                }
            }
            cur_tile = k;
        };
        // Two register sets: the patch of chunk q+3 is requested while chunk q is being consumed, i.e. every load has two
        // full chunk times (~6 us) to return before it is written to LDS -- one chunk time is less than a loaded HBM trip.
        uint4 stgA[NLD], stgB[NLD];
        auto load_chunk = [&](int q, uint4 (&stg)[NLD]) {           // global chunk index q -> (tile, chunk)
            const int k = q / p.nchunks, ch = q % p.nchunks;
            if (k != cur_tile) set_tile(k);
            const uint4* src = gin + (long long)ch * 4 * HW;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {                          // exactly NLD loads, always (vmcnt bookkeeping below)
                const uint4 v = src[soff[i] >= 0 ? soff[i] : 0];
                stg[i] = soff[i] >= 0 ? v : make_uint4(0, 0, 0, 0);
            }
        };
        const uint4* gA = reinterpret_cast<const uint4*>(p.wpk);
        int dma_k = -1, dma_mt = 0;
        auto dma_A = [&](int q) {                                     // A fragments of global chunk q -> its LDS stage
            const int kq = q / p.nchunks;
            if (kq != dma_k) {                                        // (tile-skip mode: tile_coords reads the work list)
                int x0, y0, b0;
                tile_coords(kq, dma_mt, x0, y0, b0);
                dma_k = kq;
            }
            const int mt = dma_mt;
            const uint4* src = gA + ((long long)mt * p.nchunks + q % p.nchunks) * AUNITS + ltid;
            uint4* dst = smem_u + (q & 1) * STAGE + UNITS + wn * 64;
#pragma unroll
            for (int i = 0; i < NDA; ++i)
                __builtin_amdgcn_global_load_lds((glb_void*)(src + i * 256), (lds_void*)(dst + i * 256), 16, 0, 0);
        };
        auto store_chunk = [&](int stage, const uint4 (&stg)[NLD]) {
            uint4* dst = smem_u + stage * STAGE;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int u = ltid + i * 256;
                if (u < UNITS) dst[u] = stg[i];
            }
        };
        // ---- small ACE epilogue operands of tile k: per-channel parameters, noise, label patch.  Loaded into registers at
        // the tile's second-to-last chunk, written to LDS at its last chunk.
        float4 parr = make_float4(0.f, 0.f, 0.f, 0.f);
        float nzr[2] = {0.f, 0.f};
        uint8_t labr[4] = {255, 255, 255, 255};
        auto epi_load = [&](int k) {
            int mt, x0, y0, b0;
            tile_coords(k, mt, x0, y0, b0);
            const int C = p.C;
            if (ltid < 8 * NPAR) {                                        // par[run][which]
                const int run = ltid / NPAR, which = ltid % NPAR;
                const int c0 = (mt * 8 + run) * 4;
                const float* src = which == 0 ? p.bias_g : (which == 1 ? p.bias_b : (which == 2 ? p.bn_a : (which == 3 ? p.bn_d : p.nv)));
                parr = *reinterpret_cast<const float4*>(src + (c0 < C ? c0 : 0));
                // the output scale is folded into the constants: s * (1 + bias_g), s * bias_b (ace_quad)
                const float osc = p.out_scale != 0.f ? p.out_scale : 1.f;
                if (which == 0) parr = make_float4((parr.x + 1.f) * osc, (parr.y + 1.f) * osc, (parr.z + 1.f) * osc, (parr.w + 1.f) * osc);
                if (which == 1) parr = make_float4(parr.x * osc, parr.y * osc, parr.z * osc, parr.w * osc);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = ltid + i * 256;
                const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
                const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
                const bool ok = b < p.B && y < p.H && x < p.W;
                nzr[i] = p.noise[ok ? (long long)b * p.noise_bstride + (long long)x * p.H + y : 0];
            }
            if (p.lut) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = ltid + i * 256;
                    labr[i] = 255;
                    if (e < TB * LH * LW) {
                        const int tb = e / (LH * LW), ly = (e / LW) % LH, lx = e % LW;
                        const int b = b0 + tb, y = y0 - 1 + ly, x = x0 - 1 + lx;
                        const bool in = b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                        const uint8_t v = p.lab[in ? (long long)b * HW + y * p.W + x : 0];
                        labr[i] = in ? v : (uint8_t)255;
                    }
                }
            }
        };
        auto epi_store = [&]() {
            if (ltid < 8 * NPAR) reinterpret_cast<float4*>(smem_u + PAR0)[ltid] = parr;
#pragma unroll
            for (int i = 0; i < 2; ++i) reinterpret_cast<float*>(smem_u + NZ0)[ltid + i * 256] = nzr[i];
            if (p.lut) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = ltid + i * 256;
                    if (e < TB * LH * LW) reinterpret_cast<uint8_t*>(smem_u + LAB0)[e] = labr[i];
                }
            }
        };
        // CP: boundary-pixel list + count of tile k -> LDS buffer k & 1, one tile ahead of the consumers
        [[maybe_unused]] unsigned listr = 0;
        [[maybe_unused]] int cntr = 0;
        auto list_load = [&](int k) {
            if constexpr (CP) {
                const unsigned tile = p.sp_work[first + k * (int)gridDim.x] & 0xFFFFFu;
                listr = reinterpret_cast<const unsigned*>(p.sp_list + (long long)tile * (TW * TH))[ltid];
                if (ltid == 0) cntr = p.sp_cnt[tile];
            }
        };
        auto list_store = [&](int k) {
            if constexpr (CP) {
                reinterpret_cast<unsigned*>(smem_u + LIST0 + (k & 1) * 64)[ltid] = listr;
                if (ltid == 0) reinterpret_cast<int*>(smem_u + META0)[k & 1] = cntr;
            }
        };
        if (Q > 0) {
            dma_A(0);
            load_chunk(0, stgA);
            if (Q > 1) load_chunk(1, stgB);
            store_chunk(0, stgA);
            if (Q > 2) load_chunk(2, stgA);
            list_load(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (Q > 0) list_store(0);
        __syncthreads();                                      // stage 0 ready
        // iteration q: A(q+1) by DMA and patch q+1 (requested two iterations ago) -> LDS stage (q+1)&1, request patch q+3.
        // Vector-memory results return in order, so waiting until only the NLD newest loads (the patch just requested)
        // are outstanding guarantees that the DMA has landed, without draining the patch prefetch.
        auto iter = [&](int q, uint4 (&stg)[NLD]) {
            const int k = q / p.nchunks, ch = q % p.nchunks;
            if (q + 1 < Q) {
                dma_A(q + 1);
                store_chunk((q + 1) & 1, stg);
            }
            if (pre && ch == p.nchunks - 1) {                 // consumers read these after this iteration's barrier
                epi_store();
                if (k + 1 < my_tiles) list_store(k + 1);
            }
            if (q + 3 < Q) {
                load_chunk(q + 3, stg);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (pre && ch == p.nchunks - 2) {                 // younger than everything waited on above
                epi_load(k);
                if (k + 1 < my_tiles) list_load(k + 1);
            }
            __syncthreads();
        };
        for (int q = 0; q < Q; q += 2) {
            iter(q, stgB);
            if (q + 1 < Q) iter(q + 1, stgA);
        }
        return;
    }

    // -------------------------------------------------------------------- consumers
    int ub[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
        ub[n] = (lane >> 5) * 2 * PLANE + tb * (PH * PW) + ty * PW + tx;
    }
    __syncthreads();                                          // stage 0 ready
    int q = 0;
    float amax = 0.f;                                         // running max |out * out_scale| of this wave (ACE, pass 0)
    const float extra = (pre && p.pass == 1) ? sh16_dyn_extra(*p.out_amax) : 1.f;
    // dbg bit 256 (profiling only): wave 0 of every block stamps s_memtime at tile start / end of the k-loop / end of the
    // epilogue into p.partial (3 x int64 per tile, 64 tiles per block)
    const bool stamp = (p.dbg & 256) && wn == 0 && lane == 0 && p.partial;
    long long* stamps = reinterpret_cast<long long*>(p.partial) + (long long)blockIdx.x * 64 * 3;
    for (int k = 0; k < my_tiles; ++k) {
        int mtile64, x0, y0, b0;
        tile_coords(k, mtile64, x0, y0, b0);
        if (stamp && k < 64) stamps[k * 3] = __builtin_amdgcn_s_memtime();
        [[maybe_unused]] int nsub = 4, cnt = 0;               // CP: this wave's sub-tiles wn, wn + 4, ... of the tile's compacted pixels
        [[maybe_unused]] const uint16_t* lst = nullptr;
        constexpr int sst = 4 / MG;                           // CP: this wave's sub-tiles are sw0, sw0 + sst, ...
        [[maybe_unused]] const int sw0 = wn % sst;
        if constexpr (CP) {
            cnt = reinterpret_cast<const int*>(smem_u + META0)[k & 1];
            lst = reinterpret_cast<const uint16_t*>(smem_u + LIST0 + (k & 1) * 64);
            const int NS = (cnt + 31) >> 5;
            if constexpr (PAIR) mtile64 += wn / sst;
            nsub = NS > sw0 ? (NS - sw0 + sst - 1) / sst : 0;
            if (PAIR && mtile64 >= p.mtiles) nsub = 0;        // odd number of row tiles: the last pair is half empty
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int slot = (sw0 + sst * n) * 32 + (lane & 31);
                const int idx = lst[slot < cnt ? slot : cnt - 1];     // slots beyond the count repeat the last boundary pixel
                ub[n] = (lane >> 5) * 2 * PLANE + (idx >> 5) * PW + (idx & 31);
            }
        }
        constexpr int NACC = PAIR ? 2 : 4;                    // sub-tiles a wave accumulates
        f32x16 acc[2][NACC];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NACC; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        // CP, at most two sub-tiles: plain double-buffered k-step (the hand-ordered one below is written for four sub-tiles)
        auto kloop_small = [&](auto ns_) {
            constexpr int NS_ = decltype(ns_)::value;
            for (int ch = 0; ch < p.nchunks; ++ch, ++q) {
                if constexpr (NS_ > 0) {
                    const uint4* sb = smem_u + (q & 1) * STAGE;
                    const uint4* sa = sb + UNITS + lane;
                    uint4 a_c[4], bh_c[NS_], bl_c[NS_];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a_c[i] = sa[i * 64];
#pragma unroll
                    for (int n = 0; n < NS_; ++n) {
                        bh_c[n] = sb[ub[n]];
                        bl_c[n] = sb[ub[n] + PLANE];
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        uint4 a_n[4], bh_n[NS_], bl_n[NS_];
#pragma unroll
                        for (int i = 0; i < 4; ++i) a_n[i] = a_c[i];
#pragma unroll
                        for (int n = 0; n < NS_; ++n) { bh_n[n] = bh_c[n]; bl_n[n] = bl_c[n]; }
                        if (t + 1 < NT) {
                            const int knxt = ((t + 1) / KS) * PW + ((t + 1) % KS);
#pragma unroll
                            for (int i = 0; i < 4; ++i) a_n[i] = sa[((t + 1) * 4 + i) * 64];
#pragma unroll
                            for (int n = 0; n < NS_; ++n) {
                                bh_n[n] = sb[ub[n] + knxt];
                                bl_n[n] = sb[ub[n] + knxt + PLANE];
                            }
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NS_; ++n) {
                                if (TERMS == 3) {
                                    acc[m][n] = mfma16<TERMS>(a_c[m * 2 + 1], bh_c[n], acc[m][n]);
                                    acc[m][n] = mfma16<TERMS>(a_c[m * 2 + 0], bl_c[n], acc[m][n]);
                                }
                                acc[m][n] = mfma16<TERMS>(a_c[m * 2 + 0], bh_c[n], acc[m][n]);
                            }
#pragma unroll
                        for (int i = 0; i < 4; ++i) a_c[i] = a_n[i];
#pragma unroll
                        for (int n = 0; n < NS_; ++n) { bh_c[n] = bh_n[n]; bl_c[n] = bl_n[n]; }
                    }
                }
                __syncthreads();
            }
        };
        // pair entries: A fragments of this wave's row tile straight from L2, four to six taps ahead in a 9-slot register ring (the
        // stream does not depend on the chunk barriers); B fragments from the staged patch as above
        auto kloop_direct = [&](auto ns_) {
            constexpr int NS_ = decltype(ns_)::value;
            constexpr int PD = NS_ <= 1 ? 6 : 4;               // taps of lead: one sub-tile is ~200 cycles of MFMA per tap, an L2 trip ~1.5 k
            const uint4* ga = reinterpret_cast<const uint4*>(p.wpk) + (long long)mtile64 * p.nchunks * AUNITS + lane;
            uint4 ar[NT][4];
            if constexpr (NS_ > 0) {
#pragma unroll
                for (int d = 0; d < PD; ++d)
#pragma unroll
                    for (int i = 0; i < 4; ++i) ar[d][i] = ga[(d * 4 + i) * 64];
            }
            for (int ch = 0; ch < p.nchunks; ++ch, ++q) {
                if constexpr (NS_ > 0) {
                    const uint4* sb = smem_u + (q & 1) * STAGE;
                    const uint4* gc = ga + (long long)ch * AUNITS;
                    const bool more = ch + 1 < p.nchunks;
                    uint4 bh_c[NS_], bl_c[NS_];
#pragma unroll
                    for (int n = 0; n < NS_; ++n) {
                        bh_c[n] = sb[ub[n]];
                        bl_c[n] = sb[ub[n] + PLANE];
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (t + PD < NT) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) ar[t + PD][i] = gc[((t + PD) * 4 + i) * 64];
                        } else if (more) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) ar[t + PD - NT][i] = gc[AUNITS + ((t + PD - NT) * 4 + i) * 64];
                        }
                        uint4 bh_n[NS_], bl_n[NS_];
#pragma unroll
                        for (int n = 0; n < NS_; ++n) { bh_n[n] = bh_c[n]; bl_n[n] = bl_c[n]; }
                        if (t + 1 < NT) {
                            const int knxt = ((t + 1) / KS) * PW + ((t + 1) % KS);
#pragma unroll
                            for (int n = 0; n < NS_; ++n) {
                                bh_n[n] = sb[ub[n] + knxt];
                                bl_n[n] = sb[ub[n] + knxt + PLANE];
                            }
                        }
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int n = 0; n < NS_; ++n) {
                                if (TERMS == 3) {
                                    acc[m][n] = mfma16<TERMS>(ar[t][m * 2 + 1], bh_c[n], acc[m][n]);
                                    acc[m][n] = mfma16<TERMS>(ar[t][m * 2 + 0], bl_c[n], acc[m][n]);
                                }
                                acc[m][n] = mfma16<TERMS>(ar[t][m * 2 + 0], bh_c[n], acc[m][n]);
                            }
#pragma unroll
                        for (int n = 0; n < NS_; ++n) { bh_c[n] = bh_n[n]; bl_c[n] = bl_n[n]; }
                    }
                }
                __syncthreads();
            }
        };
        bool small_done = false;
        if constexpr (CP) {
            if constexpr (PAIR) {
                if (nsub == 0) kloop_direct(std::integral_constant<int, 0>{});
                else if (nsub == 1) kloop_direct(std::integral_constant<int, 1>{});
                else kloop_direct(std::integral_constant<int, 2>{});
                small_done = true;
            } else
            if (nsub == 0) { kloop_small(std::integral_constant<int, 0>{}); small_done = true; }
            else if (nsub == 1) { kloop_small(std::integral_constant<int, 1>{}); small_done = true; }
            else if (nsub == 2) { kloop_small(std::integral_constant<int, 2>{}); small_done = true; }
        }
        if constexpr (!PAIR) {
        if (!small_done)
        for (int ch = 0; ch < p.nchunks; ++ch, ++q) {
            const uint4* sb = smem_u + (q & 1) * STAGE;
            const uint4* sa = sb + UNITS + lane;
            // One wave per SIMD: nothing hides a stalled MFMA stream, so the k-step is hand-ordered and pinned with sched_barrier.
            // A tap is two halves of 12 MFMAs: pixel sub-tiles {0,1}, then {2,3} (term-major inside a half: the same accumulator
            // comes back every 4th MFMA; tools/mfma_dep.hip: 2.6 % below 8 independent accumulators).  The B fragments are
            // reloaded IN PLACE: those of sub-tiles {2,3} for this tap during the first half, those of {0,1} for the next tap
            // during the second half, i.e. right after their last use -- no second B buffer; only the A fragments (live
            // through the whole tap) are double-buffered.
            uint4 a_cur[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a_cur[i] = sa[i * 64];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = sb[ub[n]];
                bl[n] = sb[ub[n] + PLANE];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint4 a_nxt[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a_nxt[i] = a_cur[i];
                const int kcur = (t / KS) * PW + (t % KS), knxt = ((t + 1) / KS) * PW + ((t + 1) % KS);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        // group i of this half: one operand fetch, then 2 MFMAs
                        if (half == 0) {
                            if (i == 0) bh[2] = sb[ub[2] + kcur];
                            if (i == 1) bl[2] = sb[ub[2] + kcur + PLANE];
                            if (i == 2) bh[3] = sb[ub[3] + kcur];
                            if (i == 3) bl[3] = sb[ub[3] + kcur + PLANE];
                            if (t + 1 < NT && i == 4) a_nxt[0] = sa[((t + 1) * 4 + 0) * 64];
                            if (t + 1 < NT && i == 5) a_nxt[1] = sa[((t + 1) * 4 + 1) * 64];
                        } else if (t + 1 < NT) {
                            if (i == 0) bh[0] = sb[ub[0] + knxt];
                            if (i == 1) bl[0] = sb[ub[0] + knxt + PLANE];
                            if (i == 2) bh[1] = sb[ub[1] + knxt];
                            if (i == 3) bl[1] = sb[ub[1] + knxt + PLANE];
                            if (i == 4) a_nxt[2] = sa[((t + 1) * 4 + 2) * 64];
                            if (i == 5) a_nxt[3] = sa[((t + 1) * 4 + 3) * 64];
                        }
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * i + jj, term = j >> 2, m = (j & 3) >> 1, n = half * 2 + (j & 1);
                            if (TERMS != 3 && term != 2) continue;
                            acc[m][n] = mfma16<TERMS>(a_cur[m * 2 + (term == 0 ? 1 : 0)], term == 1 ? bl[n] : bh[n], acc[m][n]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) a_cur[i] = a_nxt[i];
            }
            __syncthreads();
        }
        }
        // after the last chunk's barrier the consumers no longer touch LDS: the loaders go on staging the next tile
        // while the epilogue runs
        if (stamp && k < 64) stamps[k * 3 + 1] = __builtin_amdgcn_s_memtime();
        if (CH_ABL(p.dbg & 4)) continue;
        if constexpr (!pre) {
            sh16_epilogue<TW, TH, TB, EPI, TERMS == 2>(p, acc, mtile64, wn, lane, x0, y0, b0);
        } else {
            // ---- ACE epilogue fed from LDS (x, parameters, noise, labels): no global load except the style-LUT gathers
            const int C = p.C, Go = (C + 7) >> 3;
            const int hi = lane >> 5, col = lane & 31;
            const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
            const char* xbase = reinterpret_cast<const char*>(p.x) + (long long)b0 * (C >> 2) * xHW * 16;
            const float4* par = reinterpret_cast<const float4*>(smem_u + PAR0) + (PAIR ? (wn / (4 / MG)) * 8 * NPAR : 0);
            const float* nzs = reinterpret_cast<const float*>(smem_u + NZ0);
            const uint8_t* labs8 = reinterpret_cast<const uint8_t*>(smem_u + LAB0);
            const float slope = act_slope(p.act);
            const float st = (p.wscale ? p.wscale[(mtile64 < p.mtiles ? mtile64 : 0) * 64] : 1.f) * (p.in_scale_inv != 0.f ? p.in_scale_inv : 1.f) *
                             (p.out_scale != 0.f ? p.out_scale : 1.f);
            char* obase = reinterpret_cast<char*>(p.out) + (long long)b0 * Go * 2 * HW * 16;
            const unsigned lrs = (unsigned)p.lut_rs * 4u, lns = (unsigned)p.lut_ns * 4u;
            const char* lbase = reinterpret_cast<const char*>(p.lut) + (long long)b0 * p.lut_bs * lns;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            auto body = [&](auto resc) {
                constexpr bool RESC = decltype(resc)::value;
#pragma unroll
                for (int n = 0; n < NACC; ++n) {
                    int idx = wn * 128 + n * 32 + col;
                    if constexpr (CP) {                              // this lane's compacted boundary pixel of sub-tile wn + 4 n
                        const int slot = (sw0 + sst * n) * 32 + col;
                        if (n >= nsub || slot >= cnt) continue;
                        idx = lst[slot];
                    }
                    const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
                    const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
                    if (b >= p.B || y >= p.H || x >= p.W) continue;
                    const float nz = nzs[idx];
                    const uint8_t* lp = labs8 + tb * (LH * LW) + ty * LW + tx;      // 3x3 neighbourhood origin
                    unsigned loff[9];                 // per-tap byte offset of (sample, label) column + tap row block
                    if (p.lut) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) {
                            const unsigned j = lp[(t / 3) * LW + (t % 3)];          // >= 19 (255 outside the image) -> zero column 19
                            loff[t] = ((unsigned)tb * (unsigned)p.lut_bs + (j < 19u ? j : 19u)) * lns + (unsigned)(t * 2 * C) * lrs;
                        }
                    }
                    const unsigned oo = (((unsigned)tb * Go * 2 + hi) * (unsigned)HW + (unsigned)y * p.W + (unsigned)x) * 16u;
                    const unsigned xo = ((unsigned)tb * (unsigned)(C >> 2) * xHW + (unsigned)(y >> p.x_up) * xW + (unsigned)(x >> p.x_up)) * 16u;
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const int g = mtile64 * 4 + rq, run = rq * 2 + hi, c0 = g * 8 + 4 * hi;
                        if (g >= Go) continue;
                        const bool cok = c0 < C;
                        const unsigned cc = cok ? c0 : 0;
                        const float4 pg = par[run * NPAR + 0], pb = par[run * NPAR + 1], pa = par[run * NPAR + 2],
                                     pd = par[run * NPAR + 3], pn = par[run * NPAR + 4];
                        const float4 x4 = *reinterpret_cast<const float4*>(xbase + (xo + (cc >> 2) * (unsigned)xHW * 16u));
                        float4 sg = z4, sb = z4;
                        if (p.lut) {
#pragma unroll
                            for (int t = 0; t < 9; ++t) {
                                const unsigned o1 = loff[t] + cc * lrs;
                                const float4 g4 = *reinterpret_cast<const float4*>(lbase + o1);
                                const float4 b4 = *reinterpret_cast<const float4*>(lbase + (o1 + (unsigned)C * lrs));
                                sg.x += g4.x; sg.y += g4.y; sg.z += g4.z; sg.w += g4.w;
                                sb.x += b4.x; sb.y += b4.y; sb.z += b4.z; sb.w += b4.w;
                            }
                        }
                        const float4 bg = make_float4(pg.x + sg.x, pg.y + sg.y, pg.z + sg.z, pg.w + sg.w);
                        const float4 bb = make_float4(pb.x + sb.x, pb.y + sb.y, pb.z + sb.z, pb.w + sb.w);
                        uint4 w = ace_quad<RESC, TERMS == 2>(acc[0][n][rq * 4 + 0], acc[0][n][rq * 4 + 1], acc[0][n][rq * 4 + 2], acc[0][n][rq * 4 + 3],
                                                 acc[1][n][rq * 4 + 0], acc[1][n][rq * 4 + 1], acc[1][n][rq * 4 + 2], acc[1][n][rq * 4 + 3],
                                                 st, bg, bb, pa, pd, pn, x4, nz, slope, extra, amax);
                        if (!cok) w = make_uint4(0, 0, 0, 0);
                        *reinterpret_cast<uint4*>(obase + (oo + (unsigned)g * 2u * (unsigned)HW * 16u)) = sh16_pair_swap(w);
                    }
                }
            };
            if (p.pass == 1) body(std::true_type{});
            else body(std::false_type{});
        }
        if (stamp && k < 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (profiling only) include the store drain
            stamps[k * 3 + 2] = __builtin_amdgcn_s_memtime();
        }
    }
    if constexpr (pre) {
        if (p.pass != 1 && p.out_amax) {                       // one atomic per consumer wave per launch
            amax = sh16_wave_max(amax);
            if (lane == 0) sh16_slot_max(p.out_amax, amax);
        }
    }
}

template <int KS, int TW, int TH, int TB, int EPI, int TERMS = 3, int CP = 0>
hipError_t launch_sh16_ws(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ShCfg<KS, TW, TH, TB>;
    if (p.in_mode != IN_DIRECT) return hipErrorInvalidValue;     // input views are implemented in conv_sh16_kernel only
    if (CP && !(p.sp_work && p.sp_list && p.sp_cnt && p.sp_total)) return hipErrorInvalidValue;
    auto kern = conv_sh16_ws_kernel<KS, TW, TH, TB, EPI, TERMS, CP>;
    // 2 x (patch + A fragments) + (ACE) small epilogue operands: parameters, noise, label patch
    constexpr int V3_STAGE = Cfg::UNITS + KS * KS * 4 * 64;
    constexpr int V3_LDS = EPI == EPI_ACE ? (2 * V3_STAGE + 40 + 128) * 16 + ((TB * (TH + 2) * (TW + 2) + 15) / 16) * 16 + (CP ? (129 + (CP == 3 ? 120 : (CP == 2 ? 40 : 0))) * 16 : 0)
                                          : 2 * V3_STAGE * 16;
    // per device: a process may own handles on several GPUs (ch_api.cpp DeviceGuard)
    static bool attr_set[64] = {};
    static int ncus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           V3_LDS);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        ncus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        attr_set[dev] = true;
    }
    const int ncu = ncus[dev];
    p.nchunks = (p.Cin + 15) / 16;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    p.splitk = 1;
    const int ntiles = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    const int grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), V3_LDS, stream, p);
    return hipGetLastError();
}

// split-K reduce for the C4 layout: out = act(sum_s partial[s] + bias (+ res)), fixed summation order (reproducible)
template <int DUMMY>
__global__ __launch_bounds__(1024) void sh16_splitk_reduce_kernel(const ConvParams p) {
    const long long HW = (long long)p.H * p.W;
    const int C4n = (p.Mrows + 3) >> 2;
    const long long n = (long long)p.B * C4n * HW;       // float4 elements
    const int rW = p.W >> p.res_up, rH = p.H >> p.res_up;
    float amax = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < p.splitk; ++s) {
            const float4 t = reinterpret_cast<const float4*>(p.partial)[(long long)s * n + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int cg = (int)((i / HW) % C4n);
        {   // undo the operand scaling (sh16.h): 2^-k[row] / s_in
            const float isi = sh16_in_scale_inv(p);
            float4 s4 = make_float4(isi, isi, isi, isi);
            if (p.wscale) {
                const float4 t = *reinterpret_cast<const float4*>(p.wscale + cg * 4);
                s4 = make_float4(t.x * isi, t.y * isi, t.z * isi, t.w * isi);
            }
            v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w;
        }
        if (p.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cg * 4);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
        }
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.res) {
            const long long pix = i % HW;
            const int y = (int)(pix / p.W), x = (int)(pix % p.W);
            rv = reinterpret_cast<const float4*>(p.res)[(i / HW) * (rW * rH) + (long long)(y >> p.res_up) * rW + (x >> p.res_up)];
        }
        if (p.res_after_act) {
            v.x = apply_act(v.x, p.act) + rv.x; v.y = apply_act(v.y, p.act) + rv.y;
            v.z = apply_act(v.z, p.act) + rv.z; v.w = apply_act(v.w, p.act) + rv.w;
        } else {
            v.x = apply_act(v.x + rv.x, p.act); v.y = apply_act(v.y + rv.y, p.act);
            v.z = apply_act(v.z + rv.z, p.act); v.w = apply_act(v.w + rv.w, p.act);
        }
        reinterpret_cast<float4*>(p.out)[i] = v;
        amax = fmaxf(fmaxf(amax, fabsf(v.x)), fabsf(v.y));
        amax = fmaxf(fmaxf(amax, fabsf(v.z)), fabsf(v.w));
    }
    if (p.out_amax) sh16_block_slot_max(p.out_amax, amax * SH16_ACT_SCALE);       // uniform branch: every thread gets here
}

template <int KS, int TW, int TH, int TB, int EPI, int TERMS = 3, bool FUSE = false, bool INC4 = false, bool S2D = false,
          bool D2S = false>
hipError_t launch_sh16(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ShCfg<KS, TW, TH, TB>;
    auto kern = conv_sh16_kernel<KS, TW, TH, TB, EPI, TERMS, FUSE, INC4, S2D, D2S>;
    static bool attr_set[64] = {};                           // per device (a process may own handles on several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    p.nchunks = (p.Cin + 15) / 16;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    const int grid = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    p.splitk = 1;
    p.cps = p.nchunks;
    if (EPI == EPI_PLAIN && p.partial && grid < 192 && p.nchunks >= 8 && !FUSE) {
        // few tiles, long reduction (low-resolution 1024-channel layers, small batches): split K over more blocks
        const long long slab = (long long)p.B * ((p.Mrows + 3) / 4 * 4) * p.H * p.W;
        int sk = (512 + grid - 1) / grid;
        if (sk > p.nchunks / 2) sk = p.nchunks / 2;
        if ((long long)sk * slab > p.partial_cap) sk = (int)(p.partial_cap / slab);
        if (sk > 1) {
            p.cps = (p.nchunks + sk - 1) / sk;
            p.splitk = (p.nchunks + p.cps - 1) / p.cps;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid * p.splitk), dim3(256), Cfg::LDS_BYTES, stream, p);
    if (p.splitk > 1) {
        const long long n = (long long)p.B * ((p.Mrows + 3) / 4) * p.H * p.W;
        if (p.out_amax) {      // few, large blocks: one slot atomic per block (sh16.h)
            hipLaunchKernelGGL(sh16_splitk_reduce_kernel<0>, dim3(sh16_ew_grid(n)), dim3(SH16_EW_THREADS), 0, stream, p);
        } else {
            const int rg = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
            hipLaunchKernelGGL(sh16_splitk_reduce_kernel<0>, dim3(rg), dim3(256), 0, stream, p);
        }
    }
    return hipGetLastError();
}

template <int TERMS>
hipError_t launch_sh16_ws2(ConvParams p, int rows, hipStream_t stream);     // conv_sh16_ws2.h

// implemented in conv_inst_sh16*.hip
// ---- layer -> kernel selection, shared by the 3-term (f32-class) and 1-term (plain f16 operands) instantiation files
// true when dispatch_sh16_ace serves this layer with the wave-specialised persistent kernel -- the only ACE kernel that
// honours p.sp_work (tile-skip mode of the exact SPADE-interior reduction); sean_model.cpp asks before it sets the list
inline bool sh16_ace_uses_ws(const ConvParams& p) {
    const int rows = ((p.C + 31) / 32) * 64;
    const long long ntiles = (long long)(rows / 64) * ((p.W + 31) / 32) * ((p.H + 15) / 16) * p.B;
    if (CH_ABL(p.dbg & 2048) && p.Cin == 128 && p.W >= 32) return false;
    const bool ws_ok = p.W >= 32 && p.Cin >= 48;
    return ws_ok && ((p.dbg & 64) || (!(p.dbg & 128) && ntiles >= 512));
}

template <int TERMS>
hipError_t dispatch_sh16_ace(const ConvParams& p, hipStream_t s) {
    if (p.act > ACT_RELU) return hipErrorInvalidValue;   // the ACE epilogue implements none / leaky / relu only
    const int rows = ((p.C + 31) / 32) * 64;
    // dbg bit 64 forces the wave-specialised persistent kernel, bit 128 forbids it; default: layers with at least two
    // rounds of tiles per CU (its loaders then hide every tile's prologue behind the previous tile's epilogue)
    const long long ntiles = (long long)(rows / 64) * ((p.W + 31) / 32) * ((p.H + 15) / 16) * p.B;
    // dbg bit 2048: the experimental kernel of conv_sh16_ws2.h (epilogue pipelined into the next tile's k-loop; correct,
    // but its half-size tiles double the A-fragment traffic and the loaders cannot deliver it: DESIGN.md section 7)
#ifdef CH_ABLATE
    if ((p.dbg & 2048) && p.Cin == 128 && p.W >= 32) return launch_sh16_ws2<TERMS>(p, rows, s);
#endif
    const bool ws_ok = p.W >= 32 && p.Cin >= 48;
    if (ws_ok && ((p.dbg & 64) || (!(p.dbg & 128) && ntiles >= 512))) {
        {        // pixel-level compaction when the caller passes the per-tile lists (sean_model.cpp); every operand format since round 6
            if (p.sp_work && p.sp_list) {
                hipError_t e = launch_sh16_ws<3, 32, 16, 1, EPI_ACE, TERMS, 1>(p, rows, s);
                if (e != hipSuccess || !p.sp_work2) return e;
                ConvParams p2 = p;                                // the tiles with three or four sub-tiles: two row tiles per entry
                p2.sp_work = p.sp_work2;
                p2.sp_total = p.sp_total2;
                e = launch_sh16_ws<3, 32, 16, 1, EPI_ACE, TERMS, 2>(p2, rows, s);
                if (e != hipSuccess || !p.sp_work3) return e;
                p2.sp_work = p.sp_work3;                          // one or two sub-tiles: four row tiles per entry
                p2.sp_total = p.sp_total2 + 1;
                return launch_sh16_ws<3, 32, 16, 1, EPI_ACE, TERMS, 3>(p2, rows, s);
            }
        }
        return launch_sh16_ws<3, 32, 16, 1, EPI_ACE, TERMS>(p, rows, s);
    }
    if (p.W >= 32) return launch_sh16<3, 32, 16, 1, EPI_ACE, TERMS>(p, rows, s);
    if (p.W > 8) return launch_sh16<3, 16, 16, 2, EPI_ACE, TERMS>(p, rows, s);
    return launch_sh16<3, 8, 8, 8, EPI_ACE, TERMS>(p, rows, s);
}

template <int KS, int TERMS>
hipError_t dispatch_sh16_plain(const ConvParams& p, hipStream_t s) {
    // dbg bit 64: wave-specialised persistent kernel (measured slower than the 2-blocks-per-CU kernel for the plain
    // epilogue, whose residual loads it cannot hide; kept selectable for profiling)
    // Default: layers with 2..8 rounds of tiles per CU (the generator's 3x3 convs up to 128x128 at B = 16) run 6-12 % faster
    // on it (measured per layer, B = 16: 762 -> 672 us, 1478 -> 1330, 1454 -> 1368); with more tiles -- the 256^2 / 512^2
    // layers, HBM-heavy -- it is level or slower (1463 -> 1430, 1574 -> 1626) and the 2-blocks-per-CU kernel stays.
    const long long ntiles = (long long)((p.Mrows + 63) / 64) * ((p.W + 31) / 32) * ((p.H + 15) / 16) * p.B;
    const bool ws_auto = KS == 3 && !(p.dbg & 128) && ntiles >= 512 && ntiles <= 2048 && p.Cin >= 48 && p.in_mode == IN_DIRECT &&
                         p.pad_mode == PAD_ZERO;
    if (((p.dbg & 64) || ws_auto) && p.W >= 32 && !(p.partial && p.mtiles_hint_small) && !p.in2)
        return launch_sh16_ws<KS, 32, 16, 1, EPI_PLAIN, TERMS>(p, p.Mrows, s);
    if (p.in2) {        // 3x3 conv with a fused 1x1 second operand (ResBlock shortcut): 32x16 tiles only
        if constexpr (KS == 3) {
            if (p.W >= 32) return launch_sh16<3, 32, 16, 1, EPI_PLAIN, TERMS, true>(p, p.Mrows, s);
        }
        return hipErrorInvalidValue;
    }
    if (p.W >= 32) return launch_sh16<KS, 32, 16, 1, EPI_PLAIN, TERMS>(p, p.Mrows, s);
    if (p.W > 8) return launch_sh16<KS, 16, 16, 2, EPI_PLAIN, TERMS>(p, p.Mrows, s);
    return launch_sh16<KS, 8, 8, 8, EPI_PLAIN, TERMS>(p, p.Mrows, s);
}

// f32 C4 input, split while staged (conv_sh16_kernel<..., INC4 = true>): the BiSeNet trunk
template <int KS>
hipError_t dispatch_sh16_plain_c4(const ConvParams& p, hipStream_t s) {
    if (p.in2) return hipErrorInvalidValue;
    if (p.W >= 32) return launch_sh16<KS, 32, 16, 1, EPI_PLAIN, 3, false, true>(p, p.Mrows, s);
    if (p.W > 8) return launch_sh16<KS, 16, 16, 2, EPI_PLAIN, 3, false, true>(p, p.Mrows, s);
    return launch_sh16<KS, 8, 8, 8, EPI_PLAIN, 3, false, true>(p, p.Mrows, s);
}
hipError_t conv_sh16_plain_c4(const ConvParams& p, int KS, hipStream_t s);   // f32 C4 in -> f32 C4 out (KS 1 or 3)

// stride-2 convs over the space-to-depth view (conv_sh16_kernel<..., S2D = true>); KS = 2 (k = 3 / 4 kernels) or 1
template <int KS, bool INC4>
hipError_t dispatch_sh16_s2d(const ConvParams& p, hipStream_t s) {
    if (p.in2 || p.s2d_cr % 16 != 0 || p.Cin % p.s2d_cr != 0) return hipErrorInvalidValue;
    if (p.W >= 32) return launch_sh16<KS, 32, 16, 1, EPI_PLAIN, 3, false, INC4, true>(p, p.Mrows, s);
    if (p.W > 8) return launch_sh16<KS, 16, 16, 2, EPI_PLAIN, 3, false, INC4, true>(p, p.Mrows, s);
    return launch_sh16<KS, 8, 8, 8, EPI_PLAIN, 3, false, INC4, true>(p, p.Mrows, s);
}
hipError_t conv_sh16_d2s(const ConvParams& p, hipStream_t s);                // SH16 in, 2x2 taps, depth-to-space C4 out (W >= 32)
hipError_t conv_sh16_s2d(const ConvParams& p, int KS, hipStream_t s);        // SH16 in
hipError_t conv_sh16_s2d_c4(const ConvParams& p, int KS, hipStream_t s);     // f32 C4 in

// p.terms == 1 / 2 selects the single-term instantiations (operands rounded to f16 / bf16, f32 accumulate: BASELINE configs[4])
hipError_t conv_sh16_plain(const ConvParams& p, int KS, hipStream_t s);   // SH16 in -> f32 C4 out
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s);             // SH16 in -> SH16 out (fused ACE)
hipError_t conv_h16_plain(const ConvParams& p, int KS, hipStream_t s);    // the TERMS = 1 instantiations
hipError_t conv_h16_ace(const ConvParams& p, hipStream_t s);
hipError_t conv_bf16_plain(const ConvParams& p, int KS, hipStream_t s);   // the TERMS = 2 instantiations
hipError_t conv_bf16_ace(const ConvParams& p, hipStream_t s);

}  // namespace chk
