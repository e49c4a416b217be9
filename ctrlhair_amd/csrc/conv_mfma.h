// conv_mfma.h -- implicit-GEMM convolution for gfx950 on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32).  KS in {1,3,4}, stride 1 or 2, zero or reflection padding, optional nearest-x2 or
// zero-insertion-x2 view of the input (the latter turns ConvTranspose2d(k3,s2,p1,op1) into a plain conv).
//
//   GEMM view:  D[row][pixel] = sum_{ci,tap} A[row][(ci,tap)] * X[ci][pixel + tap]
//     A : weights, pre-packed on the host into per-lane MFMA A-fragments (pack_conv_weights, sean_model.cpp),
//         streamed straight from L2 into VGPRs as 16-byte loads (4 k-steps per load) -- never through LDS.
//     X : the un-expanded input patch (tile + halo) of CK input channels, staged once per chunk into LDS and
//         shared by the 4 waves; the im2col is only a per-tap LDS address offset (ds_read_b32, lanes along x
//         -> conflict-free banks).
//   Block = 256 threads = 4 wave64.  Each wave owns 64 GEMM rows (2 M-subtiles) x 128 pixels (4 N-subtiles):
//   8 independent 32x32 accumulators (128 AGPR/VGPR), 6 operand VGPRs per k-step -> the 64-cycle f32 MFMA is
//   the only busy pipe; 2 blocks/CU hide the staging barrier.
//
// Epilogues (fused, so modulation tensors never reach HBM):
//   EPI_PLAIN : + bias[row] (+ residual, optionally nearest-x2 addressed) (+ activation)   -> NCHW
//   EPI_ACE   : rows are (gamma | beta) pairs of the SPADE conv (normalization.py:249-257); adds the style LUT
//               gather (exact form of conv_gamma/conv_beta on the piecewise-constant style map,
//               normalization.py:117-153,172-173), the blend (:177-181), eval-BN + noise (:111-112) of x,
//               out = normalized*(1+gamma)+beta (:182), optional leaky_relu(0.2) (architecture.py:95)  -> NCHW
//   EPI_NHWC  : operands swapped (D^T), rows contiguous per pixel -> [pixel][row]; used to build the style LUT
//               (rows = (tap, gamma|beta, channel), pixels = (sample, label)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Timing ablations (skipped loads / epilogues / gathers: WRONG RESULTS) and superseded kernel versions kept for A/B measurements
// exist only in builds with -DCH_ABLATE (make ABLATE=1); the default library contains none of them and ch_set_option rejects
// their sean.dbg bits.
#ifdef CH_ABLATE
#define CH_ABL(x) (x)
#else
#define CH_ABL(x) 0
#endif
// sean.dbg bits that need a CH_ABLATE build: 1 / 2 / 4 / 8 skip loads, the MFMA loop, epilogues; 2048 conv_sh16_ws2_kernel;
// 65536 / 1048576 / 2097152 / 16777216 earlier interior-pass kernels; 4194304 no style-LUT gathers
constexpr int CH_ABLATE_DBG_MASK = 1 | 2 | 4 | 8 | 2048 | 65536 | 1048576 | 2097152 | 16777216 | 4194304;

namespace chk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_PLAIN = 0, EPI_ACE = 1, EPI_NHWC = 2 };
enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1 };
enum { IN_DIRECT = 0, IN_UP2_NEAREST = 1, IN_UP2_ZEROINS = 2 };

struct ConvParams {
    const float* in;        // [B][Cin][Hin][Win]
    const float* wpk;       // packed A fragments
    float* out;
    int B, Cin, H, W;       // H, W = OUTPUT spatial size
    int Hin, Win;           // physical input size (0 -> same as H, W)
    int pad;                // logical padding (top/left); -1 -> KS/2
    int pad_mode;           // PAD_ZERO / PAD_REFLECT
    int in_mode;            // IN_DIRECT / IN_UP2_NEAREST / IN_UP2_ZEROINS (logical input = 2x physical)
    int Mrows;              // real GEMM rows (Cout; 2*C for EPI_ACE is NOT used: see C)
    int nchunks;            // ceil(Cin / CK)
    int mtiles;             // ceil(rows / (64*WM)) (block tiles along M)
    int tiles_x, tiles_y, tiles_b;
    // EPI_PLAIN
    const float* bias;      // [Mrows] or null
    const float* res;       // [B][Mrows][H>>res_up][W>>res_up] or null
    int res_up;
    int res_after_act;      // 0: act(conv + bias + res) ; 1: act(conv + bias) + res
    int act;
    // EPI_ACE
    const float* x;         // [B][C][H>>x_up][W>>x_up]
    int x_up;
    int C;
    const float* bias_g;    // [C] blended biases
    const float* bias_b;
    const float* bn_a;      // rstd
    const float* bn_d;      // -mean*rstd
    const float* nv;        // noise_var*rstd
    const float* noise;     // plane base for this ACE; sample stride noise_bstride; layout [W][H]
    long long noise_bstride;
    const uint8_t* lab;     // [B][H][W]
    const float* lut;       // [B*19][9][2][C] or null (unstyled)
    int lut_bs;             // f16x3 path: columns per sample (20: labels 0-18 + an all-zero column 19 = outside the image)
    int lut_rs, lut_ns;     // f16x3 path: element (row = (t*2+gb)*C + c, n = b*lut_bs+j) lives at lut[row*lut_rs + n*lut_ns]
                            // (1, 18C) = rows contiguous per n;  (Npad, 4) = C4 layout written by the f16x3 LUT GEMM
    int splitk;             // EPI_PLAIN only: K split over `splitk` blocks, raw partial sums to `partial` slabs
    int cps;                // chunks per split
    float* partial;         // [splitk][B][Mrows][H*W] scratch (then splitk_reduce_kernel applies bias/res/act)
    long long partial_cap;  // floats available in `partial` (0 = split-K disabled)
    const void* in2;        // f16x3 plain 3x3 conv: optional second input (SH16, Cin2 channels) whose 1x1 conv with wpk2 is added
    const float* wpk2;      //   into the same accumulators (ResBlock shortcut conv_s folded into conv_1); null = none
    int Cin2;
    int terms;              // f16 MFMA path: 0/3 = three-term split operands (f32-class), 1 = hi halves only (f16 operands)
    // f16 MFMA path scaling (sh16.h): accumulators hold sum (w * 2^k[row]) * (x * s_in); the epilogue multiplies them by
    // wscale[row] / s_in (exact powers of two).  s_in = 1 / in_scale_inv, times the dynamic factor of the input's slot.
    const float* wscale;    // [GEMM rows] 2^-k per packed weight row (EPI_ACE: uniform per 64-row wave tile); null = 1
    float in_scale_inv;     // 1 / (first-pass scale of the SH16 input tensor(s)); 0 = 1
    float out_scale;        // EPI_ACE: first-pass scale of the SH16 output tensor; 0 = 1
    float out_mul;          // EPI_PLAIN: extra power of two applied to the conv term (the style LUT is stored pre-multiplied
                            // by the ACE output scale); 0 = 1
    const unsigned* in_amax;    // device slot with the recorded max of `in` (sh16_dyn_extra), null = static scale
    const unsigned* in2_amax;   // same for `in2`
    unsigned* out_amax;     // EPI_ACE: slot receiving max |out * out_scale| (pass 0) / deciding the rescue (pass 1)
    int pass;               // EPI_ACE: 0 = write with out_scale and record the maximum; 1 = return unless the recorded
                            // maximum left the window, else rewrite with the corrected scale
    int in_c4;              // set by callers of the INC4 instantiations of conv_sh16_kernel (documentation only: `in` is an f32
                            // tensor in the C4 layout [B][C/4][H][W][4], split into f16 pairs while it is staged)
    int s2d_cr, s2d_phase0; // S2D instantiations of conv_sh16_kernel (stride-2 convs): real input channels (Cin = phases x s2d_cr)
                            // and the first phase (0: 2x2-tap form of a k3 / k4 pad-1 kernel; 3: 1x1 stride-2 conv)
    int mtiles_hint_small;  // set by the caller when the layer has few tiles (prefer the split-K path over the persistent kernel)
    int dbg;                // perf experiments only: 1 = skip staging after chunk 0, 2 = skip the MFMA loop
    // exact SPADE-interior reduction (ace_sparse.h): boundary pixels of the level's tiles and the block tasks of this launch
    const uint16_t* sp_list;
    const int* sp_cnt;
    const unsigned* sp_work;
    const int* sp_total;
    const unsigned* sp_work2;   // f16x3 compacting kernel: second list (pair entries, ace_worklist mode 3) or null
    const int* sp_total2;       // [0] entries of sp_work2, [1] entries of sp_work3
    const unsigned* sp_work3;   // third list (quad entries) or null
    // EPI_NHWC
    int npix_valid;         // number of valid linear pixels (y*W+x < npix_valid)
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_LRELU: return v > 0.f ? v : 0.2f * v;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_TANH: return tanhf(v);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// bijective XCD-aware remap: physical block p (XCD p%8) -> logical id so that each XCD owns a contiguous
// range of logical ids (neighbouring logical ids share the input patch / L2 lines).
__device__ __forceinline__ int xcd_remap(int p, int n) {
    const int q = n >> 3, r = n & 7, xcd = p & 7, k = p >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Fused ACE epilogue of the exact-f32 SPADE convs (conv_mfma_kernel<..., EPI_ACE>, conv_ace_sparse_kernel): a wave tile holds
// gamma rows in acc[0] and beta rows in acc[1] of 32 channels; lane (hi = lane >> 5) owns channels 8 rq + 4 hi + (0..3) of
// its pixel of every sub-tile n.  normalization.py:111-112,117-153,172-187; architecture.py:95.
// Written for memory-level parallelism (round 3: cycle stamps showed the former per-element version -- 80 scalar parameter
// loads, 16 x loads and a serial 9-tap gather loop per pixel -- at 29 % (unstyled) to 44 % (styled) of a wave's lifetime):
// channel-run (rq) outer, so the five per-channel parameters are one float4 each per run; the 3x3 label neighbourhood of a
// pixel is fetched once and packed into 45 bits; the 18 style-LUT float4 of a (pixel, run) are independent loads.
constexpr int ACE_TG = 3;
template <int NN>
__device__ __forceinline__ void ace_epilogue_f32(const ConvParams& p, f32x16 (&acc)[2][NN], int mtile64, int hi, int b,
                                                 const int (&py)[NN], const int (&px)[NN], const bool (&ok)[NN]) {
    const int C = p.C, HW = p.H * p.W;
    const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
    const uint8_t* lb = p.lab + (long long)b * HW;
    const float* xb = p.x + (long long)b * C * xHW;
    float* ob = p.out + (long long)b * C * HW;
    const float* Lb = (p.lut && !CH_ABL(p.dbg & 4194304)) ? p.lut + (long long)b * 19 * 9 * 2 * C : nullptr;   // (bit: timing ablation)
    // Pixel (n) outer: the nine (label, tap) row offsets of the pixel's style-LUT gathers are computed once and shared by its
    // four channel runs (as 32-bit element offsets from the sample's LUT: <= 19*9*2*C floats); the five parameter float4 of a
    // run are re-read per pixel from L1 instead (per-tap 64-bit address arithmetic was the larger cost).
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        if (!ok[n]) continue;
        const int y = py[n], x = px[n];
        const float nzn = p.noise[(long long)b * p.noise_bstride + (long long)x * p.H + y];
        const int xon = (y >> p.x_up) * xW + (x >> p.x_up), oon = y * p.W + x;
        unsigned lo[9];
        unsigned lmask = 0;               // bit t: the tap carries a style term (inside the image, label < 19)
        if (Lb) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                const bool in = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                const unsigned jv = lb[in ? yy * p.W + xx : 0];              // unconditional load, select after
                const bool on = in && jv < 19u;
                lmask |= on ? 1u << t : 0u;
                lo[t] = ((on ? jv : 0u) * 9u + (unsigned)t) * 2u * (unsigned)C;
            }
        }
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int c0 = mtile64 * 32 + 8 * rq + 4 * hi;
            if (c0 >= C) continue;                               // C % 4 == 0: a run is valid as a whole
            float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
            if (Lb) {
                const float* Lc = Lb + c0;
                // taps in groups of ACE_TG: 2 x ACE_TG independent 16-byte gathers in flight per round (all eighteen at once
                // spilled accumulators)
#pragma unroll
                for (int tg = 0; tg < 9; tg += ACE_TG) {
                    float4 g4[ACE_TG], b4[ACE_TG];
#pragma unroll
                    for (int tt = 0; tt < ACE_TG; ++tt) {
                        const int t = tg + tt < 9 ? tg + tt : 8;
                        g4[tt] = *reinterpret_cast<const float4*>(Lc + lo[t]);
                        b4[tt] = *reinterpret_cast<const float4*>(Lc + lo[t] + C);
                    }
#pragma unroll
                    for (int tt = 0; tt < ACE_TG; ++tt) {
                        const float w = (tg + tt < 9 && ((lmask >> (tg + tt)) & 1u)) ? 1.f : 0.f;
                        sg.x += w * g4[tt].x; sg.y += w * g4[tt].y; sg.z += w * g4[tt].z; sg.w += w * g4[tt].w;
                        sb.x += w * b4[tt].x; sb.y += w * b4[tt].y; sb.z += w * b4[tt].z; sb.w += w * b4[tt].w;
                    }
                    __builtin_amdgcn_sched_barrier(0);           // keep the groups apart (register pressure)
                }
            }
            const float4 pg = *reinterpret_cast<const float4*>(p.bias_g + c0), pb = *reinterpret_cast<const float4*>(p.bias_b + c0);
            const float4 pa = *reinterpret_cast<const float4*>(p.bn_a + c0), pd = *reinterpret_cast<const float4*>(p.bn_d + c0);
            const float4 pn = *reinterpret_cast<const float4*>(p.nv + c0);
            const float* xp = xb + (long long)c0 * xHW + xon;
            const float xv[4] = {xp[0], xp[xHW], xp[2 * xHW], xp[3 * xHW]};
            const float g_[4] = {pg.x + sg.x, pg.y + sg.y, pg.z + sg.z, pg.w + sg.w};
            const float b_[4] = {pb.x + sb.x, pb.y + sb.y, pb.z + sb.z, pb.w + sb.w};
            const float a_[4] = {pa.x, pa.y, pa.z, pa.w}, d_[4] = {pd.x, pd.y, pd.z, pd.w}, n_[4] = {pn.x, pn.y, pn.z, pn.w};
            float* op = ob + (long long)c0 * HW + oon;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rq * 4 + e;
                const float gam = acc[0][n][r] + g_[e];
                const float bet = acc[1][n][r] + b_[e];
                const float nrm = a_[e] * xv[e] + n_[e] * nzn + d_[e];
                float o = nrm * (1.f + gam) + bet;
                o = apply_act(o, p.act);
                op[(long long)e * HW] = o;
            }
        }
    }
}

template <int KS, int STRIDE, int WM, int TW, int TH, int TB, int CK, int EPI>
struct ConvCfg {
    static constexpr int WN = 4 / WM;
    static constexpr int PW = (TW - 1) * STRIDE + KS;
    static constexpr int PH = (TH - 1) * STRIDE + KS;
    static constexpr int PLANE = TB * PH * PW;
    static constexpr int NPIX = TW * TH * TB;
    static constexpr int KSTEPS = KS * KS * CK / 2;
    static constexpr int NGROUPS = KSTEPS / 4;
    static constexpr int STAGE_ELEMS = CK * PLANE;
    static constexpr int NLOAD = (STAGE_ELEMS + 255) / 256;
    static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 4;
    static_assert(NPIX == 128 * WN, "pixel tile must be 128 px per N-wave");
    static_assert(KSTEPS % 4 == 0, "k-steps per chunk must be a multiple of 4");
};

// FUSE (EPI_PLAIN, KS = 3, stride 1): a second input `in2` (Cin2 channels, same spatial size, NCHW) whose 1x1 conv with `wpk2`
// (pack_A with KS = 1, CK = 16) is accumulated into the same tile after the 3x3 chunks -- the ResBlock's learned shortcut
// conv_s folded into conv_1 (architecture.py:69-96: out = conv_1(h1) + conv_s(hs)): no shortcut kernel, no write + residual
// read of its output.  Same patch geometry (the centre tap only is read), first fused chunk staged during the last 3x3 chunk.
template <int KS, int STRIDE, int WM, int TW, int TH, int TB, int CK, int EPI, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p) {
    static_assert(!FUSE || (KS == 3 && STRIDE == 1 && EPI == EPI_PLAIN && CK == 16), "FUSE: plain 3x3 stride-1 convs only");
    using Cfg = ConvCfg<KS, STRIDE, WM, TW, TH, TB, CK, EPI>;
    constexpr int WN = Cfg::WN, PW = Cfg::PW, PH = Cfg::PH, PLANE = Cfg::PLANE;
    constexpr int NG = Cfg::NGROUPS, NLOAD = Cfg::NLOAD, SE = Cfg::STAGE_ELEMS;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L % p.mtiles;
    int nt = L / p.mtiles;
    const int ks = (EPI == EPI_PLAIN && p.splitk > 1) ? nt % p.splitk : 0;     // K slice of this block
    if (EPI == EPI_PLAIN && p.splitk > 1) nt /= p.splitk;
    const int txi = nt % p.tiles_x; nt /= p.tiles_x;
    const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
    const int x0 = txi * TW, y0 = tyi * TH, b0 = nt * TB;
    const int mtile64 = mt * WM + wm;
    const int HW = p.H * p.W;
    const int c_lo = (EPI == EPI_PLAIN && p.splitk > 1) ? ks * p.cps : 0;
    const int c_hi = (EPI == EPI_PLAIN && p.splitk > 1) ? (c_lo + p.cps < p.nchunks ? c_lo + p.cps : p.nchunks) : p.nchunks;

    // per-lane LDS offsets of the window origin for the 4 N-subtiles
    int loff[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int idx = wn * 128 + n * 32 + (lane & 31);
        const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
        loff[n] = (lane >> 5) * PLANE + tb * (PH * PW) + ty * STRIDE * PW + tx * STRIDE;
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- staging: element e of the chunk patch -> (channel, sample-in-tile, row, col).  Done synchronously at
    // the top of each chunk iteration (registers are transient); the second block on the CU covers the wait.
    const int Hl = p.in_mode == IN_DIRECT ? p.Hin : 2 * p.Hin;   // logical input size
    const int Wl = p.in_mode == IN_DIRECT ? p.Win : 2 * p.Win;
    const int HWin = p.Hin * p.Win;
    auto stage = [&](int chunk, int buf) {
        // No scheduling fence in here on purpose: the compiler issues these loads early and sinks the LDS writes
        // below the chunk's MFMAs as far as registers allow, which is what overlaps staging with compute.
        float stg[NLOAD];
        const float* src = p.in + ((long long)b0 * p.Cin + (long long)chunk * CK) * HWin;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            int e = tid + i * 256;
            asm volatile("" : "+v"(e));     // keep the decode inside the chunk loop (no hoisted address regs)
            float v = 0.f;
            if (e < SE) {
                const int c = e / PLANE, rem = e % PLANE;
                const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
                int y = y0 * STRIDE - p.pad + py, x = x0 * STRIDE - p.pad + px;
                if (p.pad_mode == PAD_REFLECT) {
                    y = y < 0 ? -y : (y >= Hl ? 2 * (Hl - 1) - y : y);
                    x = x < 0 ? -x : (x >= Wl ? 2 * (Wl - 1) - x : x);
                }
                bool ok = b0 + tb < p.B && chunk * CK + c < p.Cin && (unsigned)y < (unsigned)Hl &&
                          (unsigned)x < (unsigned)Wl;
                if (p.in_mode != IN_DIRECT) {
                    if (p.in_mode == IN_UP2_ZEROINS && ((y | x) & 1)) ok = false;
                    y >>= 1;
                    x >>= 1;
                }
                if (ok) v = src[(tb * p.Cin + c) * HWin + y * p.Win + x];
            }
            stg[i] = v;
        }
        float* dst = smem + buf * SE;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int e = tid + i * 256;
            if (e < SE) dst[e] = stg[i];
        }
    };

    const float4* Ap = reinterpret_cast<const float4*>(p.wpk) +
                       ((long long)mtile64 * p.nchunks) * (NG * 2 * 64) + lane;

    // FUSE: the second operand's chunk c2 -> LDS buffer `buf` (centre region of the patch only: its 1x1 conv reads no halo)
    const int nch2 = FUSE ? (p.Cin2 + CK - 1) / CK : 0;
    auto stage2 = [&](int c2, int buf) {
        float stg[NLOAD];
        const float* src = static_cast<const float*>(p.in2) + ((long long)b0 * p.Cin2 + (long long)c2 * CK) * HWin;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            int e = tid + i * 256;
            asm volatile("" : "+v"(e));
            float v = 0.f;
            if (e < SE) {
                const int c = e / PLANE, rem = e % PLANE;
                const int tb = rem / (PH * PW), py = (rem / PW) % PH, px = rem % PW;
                const int y = y0 - 1 + py, x = x0 - 1 + px;
                const bool ok = b0 + tb < p.B && c2 * CK + c < p.Cin2 && py >= 1 && py < PH - 1 && px >= 1 && px < PW - 1 &&
                                y < p.H && x < p.W;
                if (ok) v = src[(tb * p.Cin2 + c) * HWin + y * p.W + x];
            }
            stg[i] = v;
        }
        float* dst = smem + buf * SE;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int e = tid + i * 256;
            if (e < SE) dst[e] = stg[i];
        }
    };

    stage(c_lo, 0);
    __syncthreads();

    for (int ch = c_lo; ch < c_hi; ++ch) {
        if (ch + 1 < c_hi) stage(ch + 1, (ch + 1 - c_lo) & 1);
        if constexpr (FUSE) {
            if (ch + 1 == c_hi && nch2 > 0) stage2(0, (ch + 1 - c_lo) & 1);
        }
        const float* sb = smem + ((ch - c_lo) & 1) * SE;
        const float4* Ac = Ap + (long long)ch * (NG * 2 * 64);
        float4 a0 = Ac[0], a1 = Ac[64];
        float bv[4], bn[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = sb[loff[n]];          // k-step 0: tap (0,0), channel pair 0
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 a0n = a0, a1n = a1;
            if (g + 1 < NG) {
                a0n = Ac[(g + 1) * 128];
                a1n = Ac[(g + 1) * 128 + 64];
            }
            const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
            const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                constexpr int HALF = CK / 2;
                const int s1 = g * 4 + q + 1;                     // prefetch the next k-step's B fragments
                if (s1 < Cfg::KSTEPS) {
                    const int t = s1 / HALF, cp = s1 % HALF;
                    const int koff = 2 * cp * PLANE + (t / KS) * PW + (t % KS);
                    asm volatile("" ::: "memory");                // no cross-tap CSE of LDS reads (VGPR pressure)
#pragma unroll
                    for (int n = 0; n < 4; ++n) bn[n] = sb[koff + loff[n]];
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (EPI == EPI_NHWC) {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[n], a0v[q], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[n], a1v[q], acc[1][n], 0, 0, 0);
                    } else {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[q], bv[n], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[q], bv[n], acc[1][n], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) bv[n] = bn[n];
            }
            a0 = a0n;
            a1 = a1n;
        }
        __syncthreads();
    }

    if constexpr (FUSE) {
        // fused 1x1 operand: 16 channels per chunk = 8 k-steps at the centre tap, into the same accumulators
        const float4* Ap2 = reinterpret_cast<const float4*>(p.wpk2) + ((long long)mtile64 * nch2) * (2 * 2 * 64) + lane;
        for (int c2 = 0; c2 < nch2; ++c2) {
            const int v = (c_hi - c_lo) + c2;                  // virtual chunk index: the LDS buffer parity continues
            if (c2 + 1 < nch2) stage2(c2 + 1, (v + 1) & 1);
            const float* sb = smem + (v & 1) * SE;
            const float4* Ac = Ap2 + (long long)c2 * (2 * 2 * 64);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 a0 = Ac[g * 128], a1 = Ac[g * 128 + 64];
                const float a0v[4] = {a0.x, a0.y, a0.z, a0.w};
                const float a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cp = g * 4 + q;                  // channel pair of this k-step
                    const int koff = 2 * cp * PLANE + PW + 1;  // centre tap
                    float bv[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) bv[n] = sb[koff + loff[n]];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[q], bv[n], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[q], bv[n], acc[1][n], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------------
    const int hi = lane >> 5, col = lane & 31;
    if (EPI == EPI_PLAIN) {
        // channel-run (m, rq) outer: the four bias values of a run are loaded once, not once per pixel; pixel offsets are
        // computed once per sub-tile (the per-element version spent a third of its instructions on 64-bit address arithmetic)
        const int rW = p.W >> p.res_up, rHW = rW * (p.H >> p.res_up);
        int pb_[4], pix_[4], rpix_[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
            const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
            pb_[n] = (b < p.B && y < p.H && x < p.W) ? b : -1;
            pix_[n] = y * p.W + x;
            rpix_[n] = (y >> p.res_up) * rW + (x >> p.res_up);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int row0 = mtile64 * 64 + m * 32 + 8 * rq + 4 * hi;
                if (row0 >= p.Mrows) continue;
                float bs[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && p.splitk <= 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bs[e] = row0 + e < p.Mrows ? p.bias[row0 + e] : 0.f;
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (pb_[n] < 0) continue;
                    const long long ob = ((long long)pb_[n] * p.Mrows + row0) * HW + pix_[n];
                    if (p.splitk > 1) {
                        float* pp = p.partial + (long long)ks * p.B * p.Mrows * HW + ob;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (row0 + e < p.Mrows) pp[(long long)e * HW] = acc[m][n][rq * 4 + e];
                        continue;
                    }
                    const float* rp = p.res ? p.res + ((long long)pb_[n] * p.Mrows + row0) * rHW + rpix_[n] : nullptr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (row0 + e >= p.Mrows) break;
                        float v = acc[m][n][rq * 4 + e] + bs[e];
                        const float rv = rp ? rp[(long long)e * rHW] : 0.f;
                        v = p.res_after_act ? apply_act(v, p.act) + rv : apply_act(v + rv, p.act);
                        p.out[ob + (long long)e * HW] = v;
                    }
                }
            }
    } else if (EPI == EPI_NHWC) {
        // swapped operands: D[i = pixel in subtile][j = row in M-subtile]; lane col = row, regs = pixels
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int idx = wn * 128 + n * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
                const int b = b0 + tb, y = y0 + ty, x = x0 + tx;
                const long long lin = ((long long)b * p.H + y) * p.W + x;
                if (b >= p.B || y >= p.H || x >= p.W || lin >= p.npix_valid) continue;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = mtile64 * 64 + m * 32 + col;
                    if (row < p.Mrows) p.out[lin * p.Mrows + row] = acc[m][n][r];
                }
            }
    } else {  // EPI_ACE: wave tile = 32 channels: acc[0] = gamma rows, acc[1] = beta rows
        int py[4], px[4];
        bool ok[4];
        int bq = b0;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int idx = wn * 128 + n * 32 + col;
            const int tx = idx % TW, ty = (idx / TW) % TH, tb = idx / (TW * TH);
            py[n] = y0 + ty;
            px[n] = x0 + tx;
            ok[n] = b0 + tb < p.B && py[n] < p.H && px[n] < p.W;
            if (n == 0) bq = b0 + tb;
        }
        if constexpr (TB == 1) {
            ace_epilogue_f32<4>(p, acc, mtile64, hi, b0, py, px, ok);
        } else {
            // several samples per tile (low resolutions): a wave's four sub-tiles may belong to different samples
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int idx = wn * 128 + n * 32 + col;
                const int tb = idx / (TW * TH);
                f32x16 a1[2][1] = {{acc[0][n]}, {acc[1][n]}};
                const int py1[1] = {py[n]}, px1[1] = {px[n]};
                const bool ok1[1] = {ok[n]};
                ace_epilogue_f32<1>(p, a1, mtile64, hi, b0 + tb < p.B ? b0 + tb : b0, py1, px1, ok1);
            }
            (void)bq;
        }
    }
}

// out = act(sum_s partial[s] + bias (+ res))  (NCHW; same epilogue semantics as EPI_PLAIN).  Slabs are summed in a
// fixed order: bit-reproducible run to run.
template <int DUMMY>
__global__ void splitk_reduce_kernel(const ConvParams p) {
    const long long HW = (long long)p.H * p.W, n = (long long)p.B * p.Mrows * HW;
    const int rW = p.W >> p.res_up, rH = p.H >> p.res_up;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.partial[(long long)s * n + i];
        const int row = (int)((i / HW) % p.Mrows);
        if (p.bias) v += p.bias[row];
        float rv = 0.f;
        if (p.res) {
            const long long pix = i % HW;
            const int y = (int)(pix / p.W), x = (int)(pix % p.W);
            rv = p.res[(i / HW) * (rW * rH) + (long long)(y >> p.res_up) * rW + (x >> p.res_up)];
        }
        p.out[i] = p.res_after_act ? apply_act(v, p.act) + rv : apply_act(v + rv, p.act);
    }
}

// host-side launcher for one instantiation
template <int KS, int STRIDE, int WM, int TW, int TH, int TB, int CK, int EPI, bool FUSE = false>
hipError_t launch_conv(ConvParams p, int rows, hipStream_t stream) {
    using Cfg = ConvCfg<KS, STRIDE, WM, TW, TH, TB, CK, EPI>;
    auto kern = conv_mfma_kernel<KS, STRIDE, WM, TW, TH, TB, CK, EPI, FUSE>;
    static bool attr_set[64] = {};                           // per device (a process may own handles on several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    p.nchunks = (p.Cin + CK - 1) / CK;
    if (p.Hin == 0) { p.Hin = p.H; p.Win = p.W; }
    if (p.pad < 0) p.pad = KS / 2;
    p.mtiles = (rows + 64 * WM - 1) / (64 * WM);
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_b = (p.B + TB - 1) / TB;
    int grid = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    p.splitk = 1;
    p.cps = p.nchunks;
    // (4x4 stride-2 layers -- the shape encoders -- split from four chunks on: layer 1, 32 -> 64 channels at 64 x 64, is 64 blocks otherwise)
    if (EPI == EPI_PLAIN && p.partial && grid < 192 && (p.nchunks >= 8 || (KS == 4 && STRIDE == 2 && p.nchunks >= 4)) && !FUSE) {
        // few tiles but a long reduction (low-resolution, wide layers: shape VAE, BiSeNet tail): split K so that
        // ~2 blocks per CU stream the weights concurrently; partial sums go to slabs and are reduced deterministically.
        const long long slab = (long long)p.B * p.Mrows * p.H * p.W;
        int sk = (512 + grid - 1) / grid;
        if (sk > p.nchunks / 2) sk = p.nchunks / 2;
        if ((long long)sk * slab > p.partial_cap) sk = (int)(p.partial_cap / slab);
        if (sk > 1) {
            p.cps = (p.nchunks + sk - 1) / sk;
            p.splitk = (p.nchunks + p.cps - 1) / p.cps;
            grid *= p.splitk;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Cfg::LDS_BYTES, stream, p);
    if (p.splitk > 1) {
        const long long n = (long long)p.B * p.Mrows * p.H * p.W;
        const int rg = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3(rg), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

// chunk sizes used by the packer and the launch table
constexpr int CK_KS3 = 16;
constexpr int CK_KS1 = 16;
constexpr int CK_S2 = 8;     // stride-2 variants (larger input patch per output tile)
inline int conv_ck(int ks, int stride) { return stride == 2 ? CK_S2 : 16; }

// entry points implemented in conv_inst_*.hip (tile config chosen from W and rows)
hipError_t conv_plain3(const ConvParams& p, hipStream_t s);          // rows = p.Mrows
hipError_t conv_plain1(const ConvParams& p, hipStream_t s);          // rows = p.Mrows
hipError_t conv_ace(const ConvParams& p, hipStream_t s);             // rows = 64-row tiles of 32 ch (gamma|beta)
hipError_t conv_nhwc1x1(const ConvParams& p, hipStream_t s);         // rows = p.Mrows
hipError_t conv_plain_s2(const ConvParams& p, int KS, hipStream_t s); // stride 2, KS in {1,3,4}

}  // namespace chk
