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
#include "conv_mfma.h"

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

template <int TW, int TH, int TB, int EPI>
__device__ __forceinline__ void sh16_epilogue(const ConvParams& p, f32x16 (&acc)[2][4], int mtile64, int wn, int lane,
                                              int x0, int y0, int b0, int ks = 0) {
    const int HW = p.H * p.W;
    // ---- epilogue ------------------------------------------------------------------------------------------
    // Per-channel parameters are loaded ONCE per wave as float4 (a lane's 16 rows are 4 runs of 4 consecutive
    // channels), not per pixel: the per-(pixel,row) scalar loads were ~3/4 of the epilogue's VMEM instructions.
    const int hi = lane >> 5, col = lane & 31;
    if (EPI == EPI_PLAIN) {
        float4 bias4[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int row0 = mtile64 * 64 + m * 32 + 8 * rq + 4 * hi;
                bias4[m][rq] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) {
                    if (row0 + 3 < p.Mrows) bias4[m][rq] = *reinterpret_cast<const float4*>(p.bias + row0);
                    else {
                        float t[4] = {0.f, 0.f, 0.f, 0.f};
                        for (int e = 0; e < 4; ++e) if (row0 + e < p.Mrows) t[e] = p.bias[row0 + e];
                        bias4[m][rq] = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
        // f32 outputs of these convs use the "C4" layout [B][C/4][H][W][4]: a lane's 4 consecutive rows of one pixel are
        // one float4, lanes run along x -> 512 contiguous bytes per half-wave store (and per residual load).
        const int C4n = (p.Mrows + 3) >> 2;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
            const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
            if (b >= p.B || y >= p.H || x >= p.W) continue;
            const long long pix = (long long)y * p.W + x;
            const int rW = p.W >> p.res_up, rH = p.H >> p.res_up;
            const long long rpix = (long long)(y >> p.res_up) * rW + (x >> p.res_up);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int row0 = mtile64 * 64 + m * 32 + 8 * rq + 4 * hi;
                    if (row0 < p.Mrows && p.splitk > 1) {       // split-K: raw partial sums to this slice's C4 slab
                        float4 v;
                        v.x = acc[m][n][rq * 4 + 0]; v.y = acc[m][n][rq * 4 + 1];
                        v.z = acc[m][n][rq * 4 + 2]; v.w = acc[m][n][rq * 4 + 3];
                        reinterpret_cast<float4*>(p.partial)[(((long long)ks * p.B + b) * C4n + (row0 >> 2)) * HW + pix] = v;
                    } else if (row0 < p.Mrows) {                // Mrows % 4 == 0
                        const float4 b4 = bias4[m][rq];
                        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.res) rv = reinterpret_cast<const float4*>(p.res)[((long long)b * C4n + (row0 >> 2)) * (rW * rH) + rpix];
                        float4 v;
                        v.x = acc[m][n][rq * 4 + 0] + b4.x; v.y = acc[m][n][rq * 4 + 1] + b4.y;
                        v.z = acc[m][n][rq * 4 + 2] + b4.z; v.w = acc[m][n][rq * 4 + 3] + b4.w;
                        if (p.res_after_act) {
                            v.x = apply_act(v.x, p.act) + rv.x; v.y = apply_act(v.y, p.act) + rv.y;
                            v.z = apply_act(v.z, p.act) + rv.z; v.w = apply_act(v.w, p.act) + rv.w;
                        } else {
                            v.x = apply_act(v.x + rv.x, p.act); v.y = apply_act(v.y + rv.y, p.act);
                            v.z = apply_act(v.z + rv.z, p.act); v.w = apply_act(v.w + rv.w, p.act);
                        }
                        reinterpret_cast<float4*>(p.out)[((long long)b * C4n + (row0 >> 2)) * HW + pix] = v;
                    }
                }
        }
    } else {  // EPI_ACE -> SH16 output.  Loop order: channel-run (rq) outer so that only one run's parameters
              // (5 float4) are live; pixel coordinates / noise / packed 3x3 label neighbourhoods are kept per n.
        const int C = p.C;
        const int Go = (C + 7) >> 3;
        const int xW = p.W >> p.x_up, xH = p.H >> p.x_up;
        _Float16* oh = reinterpret_cast<_Float16*>(p.out);
        int pb_[4], py_[4], px_[4];
        float nzv[4];
        unsigned long long labs[4];          // 9 neighbour labels x 5 bits (31 = outside the image)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
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
                    lv |= (unsigned long long)(in ? jv : 31u) << (5 * t);
                }
            }
            labs[n] = lv;
        }
        auto comp = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int g = mtile64 * 4 + rq;                  // output channel group (8 channels)
            const int c0 = g * 8 + 4 * hi;                   // this lane's 4 consecutive channels (C % 4 == 0)
            if (g >= Go) continue;
            const bool cok = c0 < C;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 pg = cok ? *reinterpret_cast<const float4*>(p.bias_g + c0) : z4;
            const float4 pb = cok ? *reinterpret_cast<const float4*>(p.bias_b + c0) : z4;
            const float4 pa = cok ? *reinterpret_cast<const float4*>(p.bn_a + c0) : z4;
            const float4 pd = cok ? *reinterpret_cast<const float4*>(p.bn_d + c0) : z4;
            const float4 pn = cok ? *reinterpret_cast<const float4*>(p.nv + c0) : z4;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int b = pb_[n], y = py_[n], x = px_[n];
                if (b < 0) continue;
                float4 sg = z4, sb = z4;
                if (p.lut && cok) {
                    const float* Lb = p.lut + (long long)b * 19 * 9 * (2 * C) + c0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const unsigned j = (unsigned)(labs[n] >> (5 * t)) & 31u;
                        const float w = j < 19u ? 1.f : 0.f;             // outside the image: contributes 0
                        const float* Lp = Lb + (long long)((j < 19u ? j : 0u) * 9 + t) * (2 * C);
                        const float4 g4 = *reinterpret_cast<const float4*>(Lp);
                        const float4 b4 = *reinterpret_cast<const float4*>(Lp + C);
                        sg.x += w * g4.x; sg.y += w * g4.y; sg.z += w * g4.z; sg.w += w * g4.w;
                        sb.x += w * b4.x; sb.y += w * b4.y; sb.z += w * b4.z; sb.w += w * b4.w;
                    }
                }
                const long long xpix = (long long)(y >> p.x_up) * xW + (x >> p.x_up);
                // x in the C4 layout [B][C/4][h][w][4]: this lane's 4 channels of the pixel are one float4
                const float4 x4 = reinterpret_cast<const float4*>(p.x)[((long long)b * (C >> 2) + ((cok ? c0 : 0) >> 2)) * (xW * xH) + xpix];
                half4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rq * 4 + e;
                    const float gam = acc[0][n][r] + comp(pg, e) + comp(sg, e);
                    const float bet = acc[1][n][r] + comp(pb, e) + comp(sb, e);
                    const float xv = comp(x4, e);
                    const float nrm = comp(pa, e) * xv + comp(pn, e) * nzv[n] + comp(pd, e);
                    float o = apply_act(nrm * (1.f + gam) + bet, p.act);
                    o = c0 + e < C ? o : 0.f;
                    const _Float16 h = (_Float16)o;
                    vh[e] = h;
                    vl[e] = (_Float16)(o - (float)h);
                }
                // Lane l (l < 32) holds channels 0-3 of the unit, lane l+32 channels 4-7 of the SAME pixel.  One
                // v_permlane32_swap per dword hands lane l its partner's hi half and lane l+32 its partner's lo half, so
                // each lane issues ONE 16-byte store (lanes 0-31: hi plane, 32-63: lo plane; 512 B contiguous each).
                uint2 wh = __builtin_bit_cast(uint2, vh), wl = __builtin_bit_cast(uint2, vl);
                {
                    auto r0 = __builtin_amdgcn_permlane32_swap(wh.x, wl.x, false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(wh.y, wl.y, false, false);
                    wh.x = r0[0]; wl.x = r0[1];
                    wh.y = r1[0]; wl.y = r1[1];
                }
                // lanes < 32 now hold {own hi (ch 0-3) in wh, partner hi (ch 4-7) in wl};
                // lanes >= 32 hold {partner lo (ch 0-3) in wh, own lo (ch 4-7) in wl}
                const long long unit = (((long long)b * Go + g) * 2 + hi) * HW + (long long)y * p.W + x;
                reinterpret_cast<uint4*>(oh)[unit] = make_uint4(wh.x, wh.y, wl.x, wl.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// in : SH16 [B][Cin/8][2][H][W][8]   (Cin % 16 == 0; padding channels hold zeros)
// EPI_PLAIN -> out f32 NCHW [B][Mrows][H][W] (bias / residual / act as conv_mfma)
// EPI_ACE   -> out SH16 [B][ceil(C/8)][2][H][W][8]  (the fused ACE epilogue of conv_mfma.h, re-split for the next conv)
template <int KS, int TW, int TH, int TB, int EPI>
__global__ __launch_bounds__(256, 2) void conv_sh16_kernel(const ConvParams p) {
    using Cfg = ShCfg<KS, TW, TH, TB>;
    constexpr int PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE, UNITS = Cfg::UNITS, NLOAD = Cfg::NLOAD, HALO = Cfg::HALO;
    constexpr int NT = KS * KS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
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

    int ub[4];   // per-lane LDS unit offset of the window origin (hi plane of this lane's k-half) per N-subtile
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
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
    // per-thread source offsets of its NLOAD patch units (chunk-invariant part), -1 = outside the image -> zeros.
    // Hoisted out of the chunk loop: the decode (3 div/mod per unit) was ~half of the kernel's VALU issue.
    int soff[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int u = tid + i * 256;
        soff[i] = -1;
        if (u < UNITS) {
            const int gh = u / PLANE, rem = u % PLANE;          // gh = group*2 + hl
            const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
            int y = y0 + py - HALO, x = x0 + px - HALO;
            const int b = b0 + tb;
            if (p.pad_mode == PAD_REFLECT) {      // nn.ReflectionPad2d (Zencoder, architecture.py:174)
                y = y < 0 ? -y : (y >= p.H ? 2 * (p.H - 1) - y : y);
                x = x < 0 ? -x : (x >= p.W ? 2 * (p.W - 1) - x : x);
            }
            if (b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                soff[i] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
        }
    }
    auto stage = [&](int chunk, int buf) {
        // No scheduling fence in here on purpose: the compiler issues these loads early and sinks the LDS writes
        // below the chunk's MFMAs as far as registers allow, which is what overlaps staging with compute.
        uint4 stg[NLOAD];
        const uint4* src = gin + (long long)chunk * 4 * HW;       // 2 groups x (hi, lo) planes per chunk
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) stg[i] = soff[i] >= 0 ? src[soff[i]] : make_uint4(0, 0, 0, 0);
        uint4* dst = smem_u + buf * UNITS;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int u = tid + i * 256;
            if (u < UNITS) dst[u] = stg[i];
        }
    };

    // A fragments: [mtile][chunk][tap][msub][hl][lane] units of 16 B
    const uint4* Ap = reinterpret_cast<const uint4*>(p.wpk) + ((long long)mtile64 * p.nchunks) * (NT * 4 * 64) + lane;

    if (!(p.dbg & 8)) stage(c_lo, 0);
    __syncthreads();

    for (int ch = c_lo; ch < c_hi; ++ch) {
        if (ch + 1 < c_hi && !(p.dbg & 1)) stage(ch + 1, (ch + 1 - c_lo) & 1);
        const uint4* sb = smem_u + ((ch - c_lo) & 1) * UNITS;
        if (p.dbg & 2) { __syncthreads(); continue; }
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
            const int koff = (t / KS) * PW + (t % KS);
            asm volatile("" ::: "memory");          // no cross-tap CSE of LDS reads
            uint4 bh[4], bl[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bh[n] = sb[ub[n] + koff];
                bl[n] = sb[ub[n] + koff + PLANE];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const half8 ah = __builtin_bit_cast(half8, a_cur[m * 2 + 0]);
                const half8 al = __builtin_bit_cast(half8, a_cur[m * 2 + 1]);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const half8 xh = __builtin_bit_cast(half8, bh[n]);
                    const half8 xl = __builtin_bit_cast(half8, bl[n]);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[m][n], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
        }
        __syncthreads();
    }

    if (p.dbg & 4) return;
    sh16_epilogue<TW, TH, TB, EPI>(p, acc, mtile64, wn, lane, x0, y0, b0, ks);
}

// -----------------------------------------------------------------------------------------------------------------
// GEN: the SPADE gamma/beta conv *generates its own input*.  Its input actv = relu(mlp_shared(one-hot labels))
// (normalization.py:239-242,253) is a pure function of the 3x3 label neighbourhood -- a 9-tap gather from a
// [19 labels x 9 taps][128] table -- so instead of materialising actv in HBM (134 MB per image per ACE at 512^2, written
// once and re-read by every M-tile block) each block rebuilds the 16-channel slice of its patch in LDS from the uint8
// label patch and an 11 KB table slice, splits it to f16 hi/lo in registers and writes the SH16 units straight into the
// MFMA staging buffer.  No HBM staging loads remain on this kernel's critical path (label patch: <1 KB per block).
template <int TW, int TH, int TB>
struct ShGenCfg : ShCfg<3, TW, TH, TB> {
    using Base = ShCfg<3, TW, TH, TB>;
    static constexpr int LW = Base::PW + 2, LH = Base::PH + 2;            // label patch (halo 2)
    static constexpr int SLICE = 19 * 9 * 16;                              // floats per table slice (16 channels)
    static constexpr int OFF_SLICE = Base::UNITS * 16;                     // byte offsets in LDS
    static constexpr int OFF_BIAS = OFF_SLICE + 2 * SLICE * 4;
    static constexpr int OFF_LAB = OFF_BIAS + 2 * 16 * 4;
    static constexpr int LDS_GEN = ((OFF_LAB + TB * LH * LW + 15) / 16) * 16;
};

template <int TW, int TH, int TB>
__global__ __launch_bounds__(256, 2) void conv_sh16_gen_kernel(const ConvParams p) {
    using Cfg = ShGenCfg<TW, TH, TB>;
    constexpr int KS = 3, PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE, UNITS = Cfg::UNITS;
    constexpr int NT = 9, LW = Cfg::LW, LH = Cfg::LH, SLICE = Cfg::SLICE;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
    float* s_slice = reinterpret_cast<float*>(reinterpret_cast<char*>(smem_u) + Cfg::OFF_SLICE);
    float* s_bias = reinterpret_cast<float*>(reinterpret_cast<char*>(smem_u) + Cfg::OFF_BIAS);
    uint8_t* s_lab = reinterpret_cast<uint8_t*>(smem_u) + Cfg::OFF_LAB;

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile64 = L % p.mtiles;
    int nt = L / p.mtiles;
    const int txi = nt % p.tiles_x; nt /= p.tiles_x;
    const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH, b0 = nt * TB;
    const int HW = p.H * p.W;

    int ub[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
        ub[n] = (lane >> 5) * 2 * PLANE + tb * (PH * PW) + ty * PW + tx;
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // label patch (255 = outside the image) + table slice / bias of chunk 0
    for (int e = tid; e < TB * LH * LW; e += 256) {
        const int tb = e / (LH * LW), ly = (e / LW) % LH, lx = e % LW;
        const int y = y0 - 2 + ly, x = x0 - 2 + lx, b = b0 + tb;
        const bool in = b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const uint8_t v = p.lab[in ? (long long)b * HW + y * p.W + x : 0];
        s_lab[e] = in ? v : (uint8_t)255;
    }
    // table in global: [19*9][Cin] floats; slice of chunk c: columns [16c, 16c+16)
    constexpr int SL4 = SLICE / 4;                              // float4 per slice (684)
    constexpr int NSL = (SL4 + 255) / 256;                      // float4 loads per thread (3)
    auto slice_load = [&](int chunk, float4 (&r)[NSL]) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int e = tid + i * 256, ee = e < SL4 ? e : 0;
            r[i] = *reinterpret_cast<const float4*>(p.gen_table + (long long)(ee >> 2) * p.Cin + chunk * 16 + (ee & 3) * 4);
        }
    };
    auto slice_store = [&](int buf, const float4 (&r)[NSL]) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int e = tid + i * 256;
            if (e < SL4) reinterpret_cast<float4*>(s_slice + buf * SLICE)[e] = r[i];
        }
    };
    {
        float4 r0[NSL];
        slice_load(0, r0);
        slice_store(0, r0);
        if (tid < 16) s_bias[tid] = p.gen_bias[tid];
    }
    __syncthreads();

    // build the SH16 patch of `chunk` (16 channels = 2 groups) from labels + table slice `buf`
    constexpr int NITEM = (2 * PLANE + 255) / 256;
    auto generate = [&](int buf) {
        const float* sl = s_slice + buf * SLICE;
        const float* bs = s_bias + buf * 16;
#pragma unroll 1
        for (int i = 0; i < NITEM; ++i) {
            const int it = tid + i * 256;
            if (it < 2 * PLANE) {
                const int g = it / PLANE, q = it % PLANE;
                const int tb = q / (PH * PW), py = (q / PW) % PH, px = q % PW;
                const uint8_t* lp = s_lab + tb * (LH * LW) + py * LW + px;
                float v[8];
                const bool inside = lp[LW + 1] != 255;          // the patch pixel itself lies in the image
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = bs[g * 8 + c];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int j = lp[(t / 3) * LW + (t % 3)];
                    if (j != 255) {
                        const float4* tp = reinterpret_cast<const float4*>(sl + (j * 9 + t) * 16 + g * 8);
                        const float4 a = tp[0], b = tp[1];
                        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w;
                        v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                    }
                }
                half8 vh, vl;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float o = inside ? fmaxf(v[c], 0.f) : 0.f;       // ReLU; zero padding outside the image
                    const _Float16 h = (_Float16)o;
                    vh[c] = h;
                    vl[c] = (_Float16)(o - (float)h);
                }
                smem_u[(g * 2 + 0) * PLANE + q] = __builtin_bit_cast(uint4, vh);
                smem_u[(g * 2 + 1) * PLANE + q] = __builtin_bit_cast(uint4, vl);
            }
        }
    };

    const uint4* Ap = reinterpret_cast<const uint4*>(p.wpk) + ((long long)mtile64 * p.nchunks) * (NT * 4 * 64) + lane;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        float4 rn[NSL];
        float bn = 0.f;
        const bool more = ch + 1 < p.nchunks;
        if (more) {
            slice_load(ch + 1, rn);
            bn = p.gen_bias[(ch + 1) * 16 + (tid & 15)];
        }
        if (!(p.dbg & 1) || ch == 0) generate(ch & 1);
        __syncthreads();
        if (!(p.dbg & 2)) {
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
                const int koff = (t / KS) * PW + (t % KS);
                asm volatile("" ::: "memory");
                uint4 bh[4], bl[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    bh[n] = smem_u[ub[n] + koff];
                    bl[n] = smem_u[ub[n] + koff + PLANE];
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const half8 ah = __builtin_bit_cast(half8, a_cur[m * 2 + 0]);
                    const half8 al = __builtin_bit_cast(half8, a_cur[m * 2 + 1]);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const half8 xh = __builtin_bit_cast(half8, bh[n]);
                        const half8 xl = __builtin_bit_cast(half8, bl[n]);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc[m][n], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
            }
        }
        if (more) {
            slice_store((ch + 1) & 1, rn);
            if (tid < 16) s_bias[((ch + 1) & 1) * 16 + tid] = bn;
        }
        __syncthreads();
    }
    if (p.dbg & 4) return;
    sh16_epilogue<TW, TH, TB, EPI_ACE>(p, acc, mtile64, wn, lane, x0, y0, b0);
}

template <int TW, int TH, int TB>
hipError_t launch_sh16_gen(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ShGenCfg<TW, TH, TB>;
    auto kern = conv_sh16_gen_kernel<TW, TH, TB>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_GEN);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nchunks = (p.Cin + 15) / 16;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    const int grid = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_GEN, stream, p);
    return hipGetLastError();
}

// -----------------------------------------------------------------------------------------------------------------
// v2: same arithmetic and layouts, different data movement.  ONE block per CU (4 waves, one per SIMD, up to 512 VGPRs):
// both operands of a 16-channel chunk -- the input patch AND the block's A fragments -- are moved HBM/L2 -> LDS by the
// LDS-DMA path (global_load_lds, 16 B per lane, no VGPR round trip) one whole chunk ahead into a 2-stage ring, so the
// only instructions between MFMAs are ds_read_b128 of fragments and the DMA issues; HBM latency is covered by a full
// chunk of MFMA work (9 k-steps x 24 MFMAs per wave) instead of by a second resident block.
// Out-of-image patch units are fetched from a 16-byte zero page (the DMA has no per-lane predicate for "write zero").
template <int KS, int TW, int TH, int TB>
struct ShCfg2 : ShCfg<KS, TW, TH, TB> {
    static constexpr int AUNITS = KS * KS * 4 * 64;                       // A fragments of one chunk (16-byte units)
    static constexpr int STAGE = ShCfg<KS, TW, TH, TB>::UNITS + AUNITS;   // units per ring stage
    static constexpr int LDS_BYTES2 = 2 * STAGE * 16;
    static_assert(LDS_BYTES2 <= 160 * 1024, "ring does not fit the 160 KiB LDS");
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int KS, int TW, int TH, int TB, int EPI>
__global__ __launch_bounds__(256, 1) void conv_sh16v2_kernel(const ConvParams p) {
    using Cfg = ShCfg2<KS, TW, TH, TB>;
    constexpr int PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE, UNITS = Cfg::UNITS, HALO = Cfg::HALO;
    constexpr int NT = KS * KS, AUNITS = Cfg::AUNITS, STAGE = Cfg::STAGE;
    constexpr int NPATCH = (UNITS + 63) / 64, NA = AUNITS / 64;          // wave-instructions per chunk
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mtile64 = L % p.mtiles;
    int nt = L / p.mtiles;
    const int txi = nt % p.tiles_x; nt /= p.tiles_x;
    const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH, b0 = nt * TB;
    const int HW = p.H * p.W;
    const int G = p.Cin >> 3;

    int ub[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
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
    const uint4* gA = reinterpret_cast<const uint4*>(p.wpk) + ((long long)mtile64 * p.nchunks) * AUNITS;
    const uint4* gzero = reinterpret_cast<const uint4*>(p.zeros);

    // DMA instruction k (0 .. NDMA-1) of this wave for `chunk` into ring stage `st`: wave w owns patch instructions
    // w, w+4, ... and A instructions w, w+4, ....  Source offsets are chunk-invariant and precomputed (doff).
    constexpr int NPW = (NPATCH + 3) / 4, NAW = (NA + 3) / 4, NDMA = NPW + NAW;
    int doff[NPW];      // -1: outside the image (zero page), -2: no instruction / lane inactive
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
        const int ins = wn + k * 4, u = ins * 64 + lane;
        doff[k] = -2;
        if (ins < NPATCH && u < UNITS) {
            const int gh = u / PLANE, rem = u % PLANE;
            const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
            const int y = y0 + py - HALO, x = x0 + px - HALO, b = b0 + tb;
            doff[k] = -1;
            if (b < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                doff[k] = ((b * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
        }
    }
    auto dma_one = [&](int chunk, int st, int k) {
        uint4* base = smem_u + st * STAGE;
        if (k < NPW) {
            const int ins = wn + k * 4;
            if (doff[k] != -2) {
                const uint4* src = doff[k] >= 0 ? gin + (long long)chunk * 4 * HW + doff[k] : gzero;
                __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(base + ins * 64), 16, 0, 0);
            }
        } else {
            const int ins = wn + (k - NPW) * 4;
            if (ins < NA)
                __builtin_amdgcn_global_load_lds((glb_void*)(gA + (long long)chunk * AUNITS + ins * 64 + lane),
                                                 (lds_void*)(base + UNITS + ins * 64), 16, 0, 0);
        }
    };

#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma_one(0, 0, k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch + 1 < p.nchunks && !(p.dbg & 1)) {
#pragma unroll
            for (int k = 0; k < NDMA; ++k) dma_one(ch + 1, (ch + 1) & 1, k);
        }
        if (p.dbg & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); continue; }
        const uint4* sb = smem_u + (ch & 1) * STAGE;
        const uint4* sa = sb + UNITS + lane;
        uint4 a_cur[4], bh[4], bl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = sa[q * 64];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bh[n] = sb[ub[n]];
            bl[n] = sb[ub[n] + PLANE];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // Hand-ordered software pipeline, pinned with sched_barrier: group i = {1 fragment read of k-step t+1,
            // 2 MFMAs of k-step t}; consecutive MFMAs hit different accumulators (term-major order).
            uint4 a_nxt[4], bhn[4], bln[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_nxt[q] = a_cur[q]; bhn[q] = bh[q]; bln[q] = bl[q]; }
            const int koff = ((t + 1) / KS) * PW + ((t + 1) % KS);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                if (t + 1 < NT) {
                    if (i < 4) a_nxt[i] = sa[((t + 1) * 4 + i) * 64];
                    else if ((i & 1) == 0) bhn[(i - 4) >> 1] = sb[ub[(i - 4) >> 1] + koff];
                    else bln[(i - 4) >> 1] = sb[ub[(i - 4) >> 1] + koff + PLANE];
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * i + jj, term = j >> 3, m = (j & 7) >> 2, n = j & 3;
                    const half8 ah = __builtin_bit_cast(half8, a_cur[m * 2 + 0]);
                    const half8 al = __builtin_bit_cast(half8, a_cur[m * 2 + 1]);
                    const half8 xh = __builtin_bit_cast(half8, bh[n]);
                    const half8 xl = __builtin_bit_cast(half8, bl[n]);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al : ah, term == 1 ? xl : xh, acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { a_cur[q] = a_nxt[q]; bh[q] = bhn[q]; bl[q] = bln[q]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    sh16_epilogue<TW, TH, TB, EPI>(p, acc, mtile64, wn, lane, x0, y0, b0);
}

// split-K reduce for the C4 layout: out = act(sum_s partial[s] + bias (+ res)), fixed summation order (reproducible)
template <int DUMMY>
__global__ void sh16_splitk_reduce_kernel(const ConvParams p) {
    const long long HW = (long long)p.H * p.W;
    const int C4n = (p.Mrows + 3) >> 2;
    const long long n = (long long)p.B * C4n * HW;       // float4 elements
    const int rW = p.W >> p.res_up, rH = p.H >> p.res_up;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < p.splitk; ++s) {
            const float4 t = reinterpret_cast<const float4*>(p.partial)[(long long)s * n + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        const int cg = (int)((i / HW) % C4n);
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
    }
}

template <int KS, int TW, int TH, int TB, int EPI>
hipError_t launch_sh16(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ShCfg<KS, TW, TH, TB>;
    auto kern = conv_sh16_kernel<KS, TW, TH, TB, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nchunks = (p.Cin + 15) / 16;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    const int grid = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    p.splitk = 1;
    p.cps = p.nchunks;
    if (EPI == EPI_PLAIN && p.partial && grid < 192 && p.nchunks >= 8) {
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
        const int rg = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(sh16_splitk_reduce_kernel<0>, dim3(rg), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

template <int KS, int TW, int TH, int TB, int EPI>
hipError_t launch_sh16v2(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ShCfg2<KS, TW, TH, TB>;
    auto kern = conv_sh16v2_kernel<KS, TW, TH, TB, EPI>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           Cfg::LDS_BYTES2);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.nchunks = (p.Cin + 15) / 16;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    const int grid = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES2, stream, p);
    return hipGetLastError();
}

// implemented in conv_inst_sh16*.hip
hipError_t conv_sh16_plain(const ConvParams& p, int KS, hipStream_t s);   // SH16 in -> f32 NCHW out
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s);             // SH16 in -> SH16 out (fused ACE)

}  // namespace chk
