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
    auto stage = [&](int chunk, int buf) {
        uint4 stg[NLOAD];
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            int u = tid + i * 256;
            asm volatile("" : "+v"(u));
            uint4 v = make_uint4(0, 0, 0, 0);
            if (u < UNITS) {
                const int gh = u / PLANE, rem = u % PLANE;          // gh = group*2 + hl
                const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
                const int y = y0 + py - HALO, x = x0 + px - HALO, b = b0 + tb;
                const int g = chunk * 2 + (gh >> 1);
                if (b < p.B && g < G && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                    v = gin[(((long long)b * G + g) * 2 + (gh & 1)) * HW + y * p.W + x];
            }
            stg[i] = v;
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

    stage(0, 0);
    __syncthreads();

    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch + 1 < p.nchunks) stage(ch + 1, (ch + 1) & 1);
        const uint4* sb = smem_u + (ch & 1) * UNITS;
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

    // ---- epilogue ------------------------------------------------------------------------------------------
    const int hi = lane >> 5, col = lane & 31;
    if (EPI == EPI_PLAIN) {
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
                for (int r = 0; r < 16; ++r) {
                    const int row = mtile64 * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (row < p.Mrows) {
                        float v = acc[m][n][r];
                        if (p.bias) v += p.bias[row];
                        float rv = 0.f;
                        if (p.res) rv = p.res[((long long)b * p.Mrows + row) * (rW * rH) + rpix];
                        v = p.res_after_act ? apply_act(v, p.act) + rv : apply_act(v + rv, p.act);
                        p.out[((long long)b * p.Mrows + row) * HW + pix] = v;
                    }
                }
        }
    } else {  // EPI_ACE -> SH16 output
        const int C = p.C;
        const int Go = (C + 7) >> 3;
        const int xW = p.W >> p.x_up, xH = p.H >> p.x_up;
        _Float16* oh = reinterpret_cast<_Float16*>(p.out);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
            const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
            if (b >= p.B || y >= p.H || x >= p.W) continue;
            float sg[16], sbt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sg[r] = 0.f; sbt[r] = 0.f; }
            if (p.lut) {
                const uint8_t* lb = p.lab + (long long)b * HW;
#pragma unroll 1
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                        const int j = lb[yy * p.W + xx];
                        const float* Lp = p.lut + ((long long)(b * 19 + j) * 9 + t) * (2 * C);
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int c4 = mtile64 * 32 + 8 * rq + 4 * hi;
                            if (c4 < C) {
                                const float4 g4 = *reinterpret_cast<const float4*>(Lp + c4);
                                const float4 b4 = *reinterpret_cast<const float4*>(Lp + C + c4);
                                sg[rq * 4 + 0] += g4.x; sg[rq * 4 + 1] += g4.y;
                                sg[rq * 4 + 2] += g4.z; sg[rq * 4 + 3] += g4.w;
                                sbt[rq * 4 + 0] += b4.x; sbt[rq * 4 + 1] += b4.y;
                                sbt[rq * 4 + 2] += b4.z; sbt[rq * 4 + 3] += b4.w;
                            }
                        }
                    }
                }
            }
            const float nz = p.noise[(long long)b * p.noise_bstride + (long long)x * p.H + y];
            const long long xpix = (long long)(y >> p.x_up) * xW + (x >> p.x_up);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int g = mtile64 * 4 + rq;              // output channel group (8 channels)
                if (g >= Go) continue;
                half4 vh, vl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = rq * 4 + e;
                    const int c = g * 8 + 4 * hi + e;
                    float o = 0.f;
                    if (c < C) {
                        const float gam = acc[0][n][r] + p.bias_g[c] + sg[r];
                        const float bet = acc[1][n][r] + p.bias_b[c] + sbt[r];
                        const float xv = p.x[((long long)b * C + c) * (xW * xH) + xpix];
                        const float nrm = p.bn_a[c] * xv + p.nv[c] * nz + p.bn_d[c];
                        o = apply_act(nrm * (1.f + gam) + bet, p.act);
                    }
                    const _Float16 h = (_Float16)o;
                    vh[e] = h;
                    vl[e] = (_Float16)(o - (float)h);
                }
                const long long unit = (((long long)b * Go + g) * 2) * HW + (long long)y * p.W + x;
                *reinterpret_cast<half4*>(oh + unit * 8 + 4 * hi) = vh;
                *reinterpret_cast<half4*>(oh + (unit + HW) * 8 + 4 * hi) = vl;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, p);
    return hipGetLastError();
}

// implemented in conv_inst_sh16*.hip
hipError_t conv_sh16_plain(const ConvParams& p, int KS, hipStream_t s);   // SH16 in -> f32 NCHW out
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s);             // SH16 in -> SH16 out (fused ACE)

}  // namespace chk
