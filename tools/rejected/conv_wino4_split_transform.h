// tools/rejected/conv_wino4_split_transform.h -- NOT part of the library (nothing includes it): round 5's attempt to halve the
// vector work of the F(4x4,3x3) kernels' input transform, kept so that it is not built again.  The two waves of a SIMD that share
// a tile's patches (row halves m = 0 / 1) each computed the 18 values V[i][3 m .. 3 m + 2] and exchanged them through a 40 KB area of
// LDS (ring of three stages + exchange = 160 KB), with a per-lane acknowledge word so that the owner never overwrites lines its
// partner has not read.  Correct (tools/wino4_bench.hip and tests/test_hip_wino.py passed on it), 72 instead of 144 transform
// operations per wave and k-step -- and SLOWER: the ten extra 1 KB LDS transfers per wave and k-step put the LDS pipe
// (128 bytes / clock / CU: A fragments 9 KB + patch rows 9 KB + exchange 10 KB per wave and k-step, x 8 waves, + 40 KB of DMA
// writes = ~2000 of the k-step's 2300 clocks) in front of the matrix pipe.  Sum over the ResBlock convs of a B = 16 step, same box,
// random operands: 14.85 ms (both waves transform the whole patch: the product kernel) / 16.9 ms (this file without the
// acknowledge words) / 18.3 ms (this file).  DESIGN.md section 7, round 5.
// conv_wino4.h -- 3x3 stride-1 zero-padded convolutions as Winograd F(4x4, 3x3) on the f32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32): 36 multiplies per 4x4 output tile and channel pair instead of 144 -- 2.25 per output pixel against 4 of
// F(2x2, 3x3) (conv_wino.h) and 9 of the direct evaluation.  Serves the ResBlock convs conv_0 / conv_1 of the SEAN generator
// (/root/reference/sean_codes/models/networks/architecture.py:82-91) from 32 x 32 pixels up (option "sean.wino" = 2).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        g: 3x3 kernel, d: 6x6 input patch, Y: 4x4 output tile   (Lavin & Gray, F(4x4,3x3))
//   M[xi][row][tile] = sum_ci U[xi][row][ci] * V[xi][ci][tile]            (xi = 0..35: thirty-six independent GEMMs)
//
// Arithmetic: every product and sum is an IEEE f32 operation (transforms: f32 adds / fmas with the constants 2, 4, 5, 8; U = G g G^T
// in double at ch_finalize, rounded once; contraction: the MFMA's f32 fma chain).  Unlike F(2x2,3x3), whose transforms only add,
// the F(4x4,3x3) transforms amplify rounding: measured 1e-5 .. 3.5e-5 per layer against a double-precision conv at O(1)
// activations (tests/test_winograd_model.py), 5-10x the error of the direct f32 sum itself, far inside the 1e-3 parity bound.
//
// Mapping to the hardware:
//   * Persistent 512-thread blocks (grid = #CUs), 8 waves = 2 per SIMD.  Block task = a spatial tile of 32 x 32 pixels (8 x 8 tiles of
//     4 x 4) x a row tile of 32 GEMM rows; wave w owns the 16-row half (w >> 2) for tile rows 2 (w & 3), 2 (w & 3) + 1 (16 tiles) and all
//     36 xi: 36 accumulators of 16x16 (144 registers, in the accumulator half of the wave's 256).  One k-step = 4 input channels = 36
//     MFMAs per wave.  (A wave with both halves -- 288 accumulator registers, one wave per SIMD -- was written first: hipcc keeps
//     accumulators beyond 256 in arch VGPRs and shuttles every one of them through an AGPR quad around its MFMA.)
//   * B operand: lane (n = lane & 15: tile, kk = lane >> 4: channel) owns one 6 x 6 patch of the k-step; the two waves that share the
//     tiles (one per row half, waves w and w + 4: the same SIMD) SPLIT its transform -- wave half m computes the 18 values V[i][j] with
//     j in {3 m, 3 m + 1, 3 m + 2} (18 LDS reads, six half row transforms of 6 operations, three column transforms of 12: 72 instead
//     of 144 vector operations per k-step) for the NEXT k-step while the MFMAs of this one run, leaves them in a 40 KB exchange area of
//     LDS and reads its partner's 18 after the k-step's barrier; the MFMAs on its own half come first.  (Both waves transforming the
//     whole patch, the first version: the transform cost 25 % of the kernel's time -- tools/wino4_bench.hip with it compiled out.)
//   * Both operands by LDS-DMA in 16-byte units (buffer_load_dwordx4 ... lds, counted waits): the patch of the tile -- image columns
//     x0 - 4 .. x0 + 35, rows y0 - 1 .. y0 + 32: units are aligned groups of 4 pixels, wholly inside or wholly outside the image (an
//     outside unit's offset lies beyond num_records: zeros) -- and the k-step's 18 KB A image.  Stage = 40 KB, ring of three (a ring
//     of four was measured level) + the exchange area = all 160 KB of LDS, the issue side two k-steps ahead as ONE flat sequence
//     across the block's tasks; one barrier per k-step.
//   * Epilogue: output transform in registers (100 operations per (row, tile)), bias / residual, 16-byte stores (8 lanes = one
//     128-byte line).
#pragma once
#include "conv_wino.h"

namespace chk {

namespace wino4 {
constexpr int TS = 32;                            // spatial tile (pixels), 8 x 8 output tiles of 4 x 4
constexpr int PROWS = 34, PUN = 10;               // patch: 34 rows of 10 units (40 floats: image columns x0 - 4 .. x0 + 35)
constexpr int PPL = PROWS * PUN;                  // units per channel plane (340)
constexpr int PUNITS = 4 * PPL;                   // 1360 patch units per k-step
constexpr int PSLOTS = 1408;                      // 2 rounds of 512 threads + 1 round of 384 (waves 0-5): 48 dummy slots
constexpr int AUNITS = 1152;                      // A image: 18 x 64 lanes x 16 bytes: 2 rounds of 512 + 1 round of 128 (waves 0-1)
constexpr int SUNITS = PSLOTS + AUNITS;           // 2560 units = 40 KB per stage
constexpr int NST = 3;
constexpr int XB_BYTES = 8 * 5 * 1024;            // exchange area: per wave 5 lines of [64 lanes][4 floats] (18 values per lane)
constexpr int LDS_BYTES = NST * SUNITS * 16 + XB_BYTES;      // 163 840
constexpr int ADW = AUNITS * 4;                   // A floats per (row tile, k-step)
}  // namespace wino4

struct Wino4Params {
    const float* in;        // [B][Cin][H][W]
    const float* wpk;       // pack_wino4_A image
    float* out;             // [B][Cout][H][W]
    int B, Cin, Cout, H, W; // H % 32 == 0, W % 32 == 0, Cin % 8 == 0, Cin >= 16
    const float* bias;      // [Cout] or null
    const float* res;       // [B][Cout][H >> res_up][W >> res_up] or null
    int res_up;
    // set by the launcher
    int nrt, ntx, nty, ntiles, ntasks, nks, rb, tbk;
};

// Position order of row half m: o = 0..17 its OWN positions (i = o / 3, j = o % 3 + 3 m: the B values the half's waves compute
// themselves), o = 18..35 the partner's (j = (o - 18) % 3 + 3 (1 - m)); xi = 6 i + j.
__host__ __device__ inline int wino4_pos(int m, int o) {
    const int oo = o < 18 ? o : o - 18, mm = o < 18 ? m : 1 - m;
    return 6 * (oo / 3) + oo % 3 + 3 * mm;
}
// image of (row tile rt, k-step s): [idx 0..17][lane][4 floats]; float e of idx holds fragment a = 4 idx + e = 36 m + o (the nine
// reads of a row half are contiguous):  U[xi = wino4_pos(m, o)][row = 32 rt + 16 m + (lane & 15)][ci = 4 s + (lane >> 4)],  U = G g G^T
template <class F>
std::vector<float> pack_wino4_A(int rows, int Cin, F get) {
    static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nrt = (rows + 31) / 32, nks = Cin / 4;
    std::vector<float> dst((size_t)nrt * nks * wino4::ADW, 0.f);
    for (int rt = 0; rt < nrt; ++rt)
        for (int s = 0; s < nks; ++s) {
            float* img = dst.data() + ((size_t)rt * nks + s) * wino4::ADW;
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = rt * 32 + m * 16 + (lane & 15), ci = 4 * s + (lane >> 4);
                    if (row >= rows) continue;
                    double g[3][3], t[6][3];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) g[a][b] = get(row, ci, a * 3 + b);
                    for (int i = 0; i < 6; ++i)
                        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
                    for (int o = 0; o < 36; ++o) {
                        const int xi = wino4_pos(m, o), i = xi / 6, j = xi % 6;
                        const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                        const int a = m * 36 + o;
                        img[((a >> 2) * 64 + lane) * 4 + (a & 3)] = (float)u;
                    }
                }
        }
    return dst;
}

// one-dimensional input transform  (B^T d):  rows of B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void wino4_in1d(float d0, float d1, float d2, float d3, float d4, float d5, float& r0, float& r1, float& r2, float& r3,
                                           float& r4, float& r5) {
    const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(-4.f, d1, d3);       // d4 - 4 d2,  d3 - 4 d1
    const float c = d4 - d2, t = d3 - d1;
    r0 = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
    r1 = a + b;
    r2 = a - b;
    r3 = __builtin_fmaf(2.f, t, c);
    r4 = __builtin_fmaf(-2.f, t, c);
    r5 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
}
// the three outputs 3 MH .. 3 MH + 2 of the same transform
template <int MH>
__device__ __forceinline__ void wino4_in1d_half(float d0, float d1, float d2, float d3, float d4, float d5, float& o0, float& o1, float& o2) {
    if constexpr (MH == 0) {
        const float a = __builtin_fmaf(-4.f, d2, d4), b = __builtin_fmaf(-4.f, d1, d3);
        o0 = __builtin_fmaf(4.f, d0, __builtin_fmaf(-5.f, d2, d4));
        o1 = a + b;
        o2 = a - b;
    } else {
        const float c = d4 - d2, t = d3 - d1;
        o0 = __builtin_fmaf(2.f, t, c);
        o1 = __builtin_fmaf(-2.f, t, c);
        o2 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    }
}
// one-dimensional output transform  (A^T m):  rows of A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void wino4_out1d(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = __builtin_fmaf(2.f, d2, d1);
    y2 = __builtin_fmaf(4.f, s2, s1);
    y3 = __builtin_fmaf(8.f, d2, d1) + m5;
}

// ================================================================================================================================
// SPADE gamma/beta conv (normalization.py:249-257) + the style convs conv_gamma / conv_beta (:117-153,172-173) + the fused ACE
// epilogue (:111-112,177-187; architecture.py:95) as F(4x4,3x3) over EVERY tile of a level -- for the levels where (nearly) every
// tile holds a boundary pixel anyway (64 x 64 and below: sean_model.cpp), where the gather kernel of conv_wino.h runs 64 products
// per 4 x 4 pixels and this one 36.  Same machinery as wino4_plain_kernel; what differs:
//   * input = the padded hidden-activation planes of conv_wino.h (WINO_AXOFF: the image sits 32 columns into rows of W + 64
//     floats, zeros left and right of it), K = 128 hidden channels (+ 20 one-hot planes: five style k-steps whose A images come
//     from a per-SAMPLE buffer, wino4_style_pack; a sixth, all-zero image makes the k-step count even);
//   * GEMM rows: a row tile = 16 channels; row r of the 16-row half m is (channel 16 rt + 8 m + 2 (r >> 2) + (r & 1), gamma | beta
//     = (r >> 1) & 1), so that the four accumulator rows of a lane are gamma and beta of TWO channels;
struct Wino4AceParams {
    const float* actv;      // [B][K][H][wino_apitch(W)]: K = 128 (+ 20 one-hot planes when wsty is set)
    const float* wpk;       // pack_wino4_A image of the SPADE rows (wino4_ace_row), 32 k-steps per row tile
    const float* wsty;      // [B][nrt][6][wino4::ADW] per-sample style images (the sixth all zero), or null (unstyled ACE)
    float* out;             // [B][C][H][W]
    const float* x;         // [B][C][H >> x_up][W >> x_up]
    int x_up, act;
    int B, C, H, W;         // H % 32 == 0, W % 32 == 0, C % 2 == 0
    const float *bias_g, *bias_b, *bn_a, *bn_d, *nv;
    const float* noise;     // plane base of this ACE, sample stride noise_bstride, layout [W][H]
    long long noise_bstride;
    int nrt, ntx, nty, ntiles, ntasks, nks, rb, tbk;      // set by the launcher
};
// GEMM row R of the packed SPADE image -> (channel, beta)
__host__ __device__ inline void wino4_ace_row(int R, int& ch, int& beta) {
    const int rt = R >> 5, m = (R >> 4) & 1, r = R & 15;
    ch = rt * 16 + m * 8 + (r >> 2) * 2 + (r & 1);
    beta = (r >> 1) & 1;
}

// ---- the block program, shared by the plain conv and the ACE conv (ACE = the parameter struct has the ACE fields); MH = the row half
//      of the calling wave: the two halves run their own copies of the code (their shares of the B transform differ), the barriers of
//      the copies pair up one to one ------------------------------------------------------------------------------------------------
template <int MH, bool ACE, class P>
__device__ __forceinline__ void wino4_body(const P& p, float* smem) {
    using namespace wino4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int tg = wave & 3;                       // tile group: tile rows 2 tg, 2 tg + 1
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;
    const int HW = p.H * p.W;
    constexpr unsigned SB = SUNITS * 16u, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;
    // input planes: the plain conv reads [Cin][H][W]; the ACE conv the padded hidden-activation planes (conv_wino.h WINO_AXOFF)
    int IPW, IPL, IXO;
    if constexpr (ACE) { IPW = wino_apitch(p.W); IPL = p.H * IPW; IXO = WINO_AXOFF; }
    else { IPW = p.W; IPL = HW; IXO = 0; }

    // task L -> (row tile, spatial tile), as conv_wino.h wino_task: 32 consecutive tasks share A images / patches through the XCD's L2
    auto task_of = [&](int L, int& rt, int& tile) {
        const int per = p.tbk * p.nrt;
        const int tgr = L / per;
        int r = L - tgr * per;
        const int tgsz = min(p.tbk, p.ntiles - tgr * p.tbk);
        const int rg = r / (tgsz * p.rb);
        r -= rg * tgsz * p.rb;
        const int rgsz = min(p.rb, p.nrt - rg * p.rb);
        const int tl = r / rgsz;
        rt = rg * p.rb + (r - tl * rgsz);
        tile = tgr * p.tbk + tl;
    };

    // ---- issue side ------------------------------------------------------------------------------------------------------------
    // patch unit u = tid + 512 i (i = 0, 1; i = 2: waves 0-5): plane u / 340, patch row (u % 340) / 10, unit column (u % 340) % 10
    unsigned voff[3];
    const unsigned va = (unsigned)tid * 16u;
    int it = lb, is = 0;
    wino_u32x4 d_in, d_a, d_s;
    unsigned so_in = 0, so_a = 0;
    auto issue_task = [&]() {
        int irt, tile;
        task_of(it, irt, tile);
        const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, ib = tile / (p.ntx * p.nty);
        const int y0 = ty * TS - 1, x0 = tx * TS - 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = i * 512 + tid;
            const int k4 = u / PPL, rem = u - k4 * PPL;
            const int py = rem / PUN, ux = rem - py * PUN;
            const int y = y0 + py, x = x0 + 4 * ux;
            // (ACE: the planes carry zero columns left and right of the image; plain: a unit outside the image reads zeros)
            const bool ok = u < PUNITS && (unsigned)y < (unsigned)p.H && (ACE || (unsigned)x < (unsigned)p.W);
            voff[i] = ok ? (unsigned)(k4 * IPL + y * IPW + x + IXO) * 4u : 0x80000000u;
        }
        if constexpr (ACE) {
            const int K = 128 + (p.wsty ? 20 : 0);
            d_in = wino_rsrc(p.actv + (long long)ib * K * IPL, (unsigned)K * IPL * 4u);
            d_a = wino_rsrc(p.wpk + (long long)irt * 32 * ADW, 32u * (unsigned)ADW * 4u);
            if (p.wsty) d_s = wino_rsrc(p.wsty + ((long long)ib * p.nrt + irt) * 6 * ADW, 6u * (unsigned)ADW * 4u);
        } else {
            d_in = wino_rsrc(p.in + (long long)ib * p.Cin * IPL, (unsigned)p.Cin * IPL * 4u);
            d_a = wino_rsrc(p.wpk + (long long)irt * p.nks * ADW, (unsigned)p.nks * ADW * 4u);
        }
        so_in = 0;
        so_a = 0;
    };
    issue_task();
    unsigned islot = lds0;
    // pieces 0, 1: patch rounds; 2, 3: A rounds (straight-line, spread over the MFMA groups); issue_tail: the third patch round (waves
    // 0-5) and the third A round (waves 0-1) + advance -- the only branches of the issue side, once per k-step behind the last group
    auto issue_piece = [&](auto pt) {
        constexpr int pc = decltype(pt)::value;
        const unsigned wb = islot + (unsigned)wave * 1024u;
        if constexpr (pc < 2) wino_dma16(voff[pc], d_in, so_in, wb + (unsigned)pc * 8192u);
        else wino_dma16(va, d_a, so_a + (unsigned)(pc - 2) * 8192u, wb + PSLOTS * 16u + (unsigned)(pc - 2) * 8192u);
    };
    auto issue_tail = [&]() {
        const unsigned wb = islot + (unsigned)wave * 1024u;
        if (wave < 6) wino_dma16(voff[2], d_in, so_in, wb + 2u * 8192u);
        if (wave < 2) wino_dma16(va, d_a, so_a + 2u * 8192u, wb + PSLOTS * 16u + 2u * 8192u);
        islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
        so_in += 16u * (unsigned)IPL;
        so_a += (unsigned)ADW * 4u;
        ++is;
        if constexpr (ACE) {
            if (is == 32 && nk > 32) {     // the style images of the task's sample follow the hidden channels
                d_a = d_s;
                so_a = 0;
            }
        }
        if (is == nk) {
            if (it + G < p.ntasks) {
                it += G;
                is = 0;
                issue_task();
            } else {                   // past the end: keep re-issuing the last k-step (never read; keeps the vmcnt counting uniform)
                is = nk - 1;
                so_in -= 16u * (unsigned)IPL;
                so_a -= (unsigned)ADW * 4u;
            }
        }
    };
    auto issue_kstep = [&]() {
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{});
        issue_tail();
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[36];                                 // position order of this half: wino4_pos(MH, o)
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const int boff = kk * (PPL * 4) + (4 * tyl) * (PUN * 4) + 4 * tx + 3;      // this lane's patch origin (floats) inside a stage
    auto stage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    auto load_row = [&](const float* sp, int r, float (&d)[6]) {               // patch row r of the lane's tile: 1 + 4 + 1 floats
        const float* q = sp + boff + r * (PUN * 4);
        d[0] = q[0];
        const f32x4 mid = *reinterpret_cast<const f32x4*>(q + 1);
        d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w;
        d[5] = q[5];
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(stage(slot) + PSLOTS * 4) + 9 * MH * 64 + lane; };
    // exchange area: line c of wave w at ((w * 5 + c) * 64 + lane) * 4 floats behind the ring
    f32x4* xb_own = reinterpret_cast<f32x4*>(smem + NST * SUNITS * 4) + (wave * 5) * 64 + lane;
    const f32x4* xb_par = reinterpret_cast<const f32x4*>(smem + NST * SUNITS * 4) + ((wave ^ 4) * 5) * 64 + lane;
    // (line 4 carries two values per lane; its third float is the lane's ACK word: the READER of the lines stores the number of the
    //  k-step whose values it has taken, the owner waits for it before overwriting them -- the two waves meet at a barrier once per
    //  k-step, but the owner's write at the END of k-step q and the partner's reads at its TOP have no barrier between them)
    volatile int* ack_own = reinterpret_cast<volatile int*>(xb_own + 4 * 64) + 2;
    volatile int* ack_par = reinterpret_cast<volatile int*>(const_cast<f32x4*>(xb_par) + 4 * 64) + 2;
    *ack_own = 0;
    int kc = 0;                                    // number of the running k-step (1, 2, ... over the block's whole task list)
    auto xb_write = [&](const float (&vo)[18]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xb_own[c * 64] = (f32x4){vo[4 * c], vo[4 * c + 1], vo[4 * c + 2], vo[4 * c + 3]};
        *reinterpret_cast<float2*>(xb_own + 4 * 64) = make_float2(vo[16], vo[17]);
    };
    auto xb_read = [&](const f32x4* xb, float (&vo)[18]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 t = xb[c * 64];
            vo[4 * c] = t.x; vo[4 * c + 1] = t.y; vo[4 * c + 2] = t.z; vo[4 * c + 3] = t.w;
        }
        const float2 t = *reinterpret_cast<const float2*>(xb + 4 * 64);
        vo[16] = t.x; vo[17] = t.y;
    };
    // this half's 18 values of the k-step staged in `slot` (patch rows -> half row transforms -> column transforms, in place)
    auto own_half = [&](unsigned slot, float (&vo)[18]) {
        const float* sp = stage(slot);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float d[6];
            load_row(sp, r, d);
            wino4_in1d_half<MH>(d[0], d[1], d[2], d[3], d[4], d[5], vo[3 * r], vo[3 * r + 1], vo[3 * r + 2]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) wino4_in1d(vo[j], vo[3 + j], vo[6 + j], vo[9 + j], vo[12 + j], vo[15 + j], vo[j], vo[3 + j], vo[6 + j], vo[9 + j], vo[12 + j], vo[15 + j]);
    };

    issue_kstep();
    issue_kstep();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned rslot = lds0;
    // one k-step: nine groups of four MFMAs -- the half's own B values `vc` first, the partner's (read from the exchange area behind the
    // barrier) after them; the half's share of the next k-step's B values goes into `vx` and, behind the last group, into the exchange area
    auto kstep = [&](float (&vc)[18], float (&vx)[18]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++kc;
        float vp[18];
        xb_read(xb_par, vp);
        int ack = 0;
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const float* spn = stage(nslot);               // (k-step q + 1 landed with q: the wait above covers everything issued)
        f32x4 F[2];
        F[0] = ap[0];
        float d[6];
        auto bval = [&](auto ot) -> float {
            constexpr int o = decltype(ot)::value;
            if constexpr (o < 18) return vc[o];
            else return vp[o - 18];
        };
        auto group = [&](auto gt) {
            constexpr int g = decltype(gt)::value;      // positions o = 4 g .. 4 g + 3
            if constexpr (g + 1 < 9) F[(g + 1) & 1] = ap[(g + 1) * 64];
            if constexpr (g < 6) load_row(spn, g, d);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 c = F[g & 1];
            acc[4 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, bval(WInt<4 * g>{}), acc[4 * g], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, bval(WInt<4 * g + 1>{}), acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, bval(WInt<4 * g + 2>{}), acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, bval(WInt<4 * g + 3>{}), acc[4 * g + 3], 0, 0, 0);
            if constexpr (g < 6)                        // half row transform of patch row g (behind the MFMAs: its LDS reads land meanwhile)
                wino4_in1d_half<MH>(d[0], d[1], d[2], d[3], d[4], d[5], vx[3 * g], vx[3 * g + 1], vx[3 * g + 2]);
            if constexpr (g >= 6) {                     // column transform g - 6, in place
                constexpr int j = g - 6;
                wino4_in1d(vx[j], vx[3 + j], vx[6 + j], vx[9 + j], vx[12 + j], vx[15 + j], vx[j], vx[3 + j], vx[6 + j], vx[9 + j], vx[12 + j], vx[15 + j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g >= 2 && g < 6) issue_piece(WInt<g - 2>{});
            if constexpr (g == 2) {                     // the partner's lines are in registers (LDS returns in order; made explicit): tell it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                *ack_par = kc;
            }
            if constexpr (g == 7) ack = *ack_own;       // (written ~1500 cycles ago; read early, checked behind the last group)
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
        group(WInt<6>{}); group(WInt<7>{}); group(WInt<8>{});
        while (ack != kc) ack = *ack_own;               // (never taken in practice)
        xb_write(vx);
        issue_tail();
        rslot = nslot;
    };
    // position o of M[i][j] in this half's accumulator order
    auto M = [&](auto it_, auto jt_, int e) -> float {
        constexpr int i = decltype(it_)::value, j = decltype(jt_)::value;
        constexpr int o = (j / 3 == MH) ? 3 * i + (j - 3 * MH) : 18 + 3 * i + (j - 3 * (1 - MH));
        return acc[o][e];
    };
    auto out_cols = [&](int e, float (&t)[4][6]) {      // A^T M: rows 0..3, columns 0..5 of accumulator row e
        auto col = [&](auto jt_) {
            constexpr int j = decltype(jt_)::value;
            wino4_out1d(M(WInt<0>{}, jt_, e), M(WInt<1>{}, jt_, e), M(WInt<2>{}, jt_, e), M(WInt<3>{}, jt_, e), M(WInt<4>{}, jt_, e), M(WInt<5>{}, jt_, e),
                        t[0][j], t[1][j], t[2][j], t[3][j]);
        };
        col(WInt<0>{}); col(WInt<1>{}); col(WInt<2>{}); col(WInt<3>{}); col(WInt<4>{}); col(WInt<5>{});
    };

    float v[18], w[18];
    own_half(lds0, v);
    xb_write(v);
    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        for (int cs = 0; cs < nk; cs += 2) {
            kstep(v, w);          // (nks is even: the launchers)
            kstep(w, v);
        }
        int crt, tile;
        task_of(ct, crt, tile);
        const int ttx = tile % p.ntx, tty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
        const int y = tty * TS + 4 * tyl, x = ttx * TS + 4 * tx;
        if constexpr (!ACE) {
            // ---- epilogue of the plain conv: bias / residual, one (row, tile) at a time (the accumulators leave little room) ----------
            const int rW = p.W >> p.res_up, rHW = rW * (p.H >> p.res_up);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = crt * 32 + MH * 16 + 4 * kk + i, rc = row < p.Cout ? row : p.Cout - 1;
                const float bsv = p.bias ? p.bias[rc] : 0.f;
                f32x4 rr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rr[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    const float* rp = p.res + ((long long)b * p.Cout + rc) * rHW;
                    if (p.res_up) {
#pragma unroll
                        for (int r2 = 0; r2 < 2; ++r2) {
                            const float2 q2 = *reinterpret_cast<const float2*>(rp + ((y >> 1) + r2) * rW + (x >> 1));
                            rr[2 * r2] = rr[2 * r2 + 1] = (f32x4){q2.x, q2.x, q2.y, q2.y};
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) rr[r] = *reinterpret_cast<const f32x4*>(rp + (y + r) * rW + x);
                    }
                }
                float t[4][6];
                out_cols(i, t);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float o0, o1, o2, o3;
                    wino4_out1d(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], o0, o1, o2, o3);
                    const f32x4 o = {o0 + bsv + rr[r].x, o1 + bsv + rr[r].y, o2 + bsv + rr[r].z, o3 + bsv + rr[r].w};
                    if (row < p.Cout) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Cout + row) * HW + (y + r) * p.W + x) = o;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ---- ACE epilogue: this lane = tile (tyl, tx) x channels cA, cA + 1: accumulator rows e (gamma) and 2 + e (beta) ----------
            const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
            f32x4 nz[4];                              // nz[c] = noise of column x + c, rows y .. y + 3 (plane layout [W][H])
            const float* nzp = p.noise + (long long)b * p.noise_bstride + (long long)x * p.H + y;
#pragma unroll
            for (int c = 0; c < 4; ++c) nz[c] = *reinterpret_cast<const f32x4*>(nzp + (long long)c * p.H);
            const int cA = crt * 16 + MH * 8 + 2 * kk;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = cA + e, cc = ch < p.C ? ch : p.C - 1;
                const float gb = 1.f + p.bias_g[cc], bb = p.bias_b[cc], pa = p.bn_a[cc], pd = p.bn_d[cc], pn = p.nv[cc];
                f32x4 xr[4];                          // x rows y .. y + 3, columns x .. x + 3
                const float* xp = p.x + ((long long)b * p.C + cc) * xHW;
                if (p.x_up) {
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        const float2 q2 = *reinterpret_cast<const float2*>(xp + ((y >> 1) + r2) * xW + (x >> 1));
                        xr[2 * r2] = xr[2 * r2 + 1] = (f32x4){q2.x, q2.x, q2.y, q2.y};
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[r] = *reinterpret_cast<const f32x4*>(xp + (y + r) * xW + x);
                }
                float tg_[4][6], tb_[4][6];
                out_cols(e, tg_);
                out_cols(2 + e, tb_);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float g0, g1, g2, g3, e0, e1, e2, e3;
                    wino4_out1d(tg_[r][0], tg_[r][1], tg_[r][2], tg_[r][3], tg_[r][4], tg_[r][5], g0, g1, g2, g3);
                    wino4_out1d(tb_[r][0], tb_[r][1], tb_[r][2], tb_[r][3], tb_[r][4], tb_[r][5], e0, e1, e2, e3);
                    const float nr[4] = {nz[0][r], nz[1][r], nz[2][r], nz[3][r]};
                    float o0 = (pa * xr[r].x + pn * nr[0] + pd) * (gb + g0) + (bb + e0);
                    float o1 = (pa * xr[r].y + pn * nr[1] + pd) * (gb + g1) + (bb + e1);
                    float o2 = (pa * xr[r].z + pn * nr[2] + pd) * (gb + g2) + (bb + e2);
                    float o3 = (pa * xr[r].w + pn * nr[3] + pd) * (gb + g3) + (bb + e3);
                    if (p.act != ACT_NONE) {
                        o0 = apply_act(o0, p.act); o1 = apply_act(o1, p.act);
                        o2 = apply_act(o2, p.act); o3 = apply_act(o3, p.act);
                    }
                    if (ch < p.C) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.C + ch) * HW + (y + r) * p.W + x) = (f32x4){o0, o1, o2, o3};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int x2 = 0; x2 < 36; ++x2) acc[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (ACE) {          // the modulation needs the registers: the half's B values of the next k-step come back from its exchange lines
            xb_read(xb_own, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's loads / stores share the counter with the ring: drain once per task
    }
}

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino4_plain_kernel(const Wino4Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) wino4_body<0, false>(p, smem);
    else wino4_body<1, false>(p, smem);
}

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino4_ace_kernel(const Wino4AceParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) wino4_body<0, true>(p, smem);
    else wino4_body<1, true>(p, smem);
}

inline bool wino4_supported(int H, int W, int Cin) { return H % wino4::TS == 0 && W % wino4::TS == 0 && Cin % 8 == 0 && Cin >= 16; }
inline void wino4_fill_launch(Wino4Params& p) {
    p.nrt = (p.Cout + 31) / 32;
    p.ntx = p.W / wino4::TS;
    p.nty = p.H / wino4::TS;
    p.ntiles = p.B * p.ntx * p.nty;
    p.ntasks = p.ntiles * p.nrt;
    p.nks = p.Cin / 4;
    p.rb = p.nrt >= 4 ? 4 : p.nrt;
    p.tbk = 32 / p.rb;
}
hipError_t conv_wino4_plain(Wino4Params p, hipStream_t s);      // conv_inst_wino4.hip
inline bool wino4_ace_supported(int H, int W, int C) { return H % wino4::TS == 0 && W % wino4::TS == 0 && C % 2 == 0; }
hipError_t conv_wino4_ace(Wino4AceParams p, hipStream_t s);
// wsty[b][rt][6][ADW] <- F(4x4,3x3) transform (G P G^T, f32) of the style LUT lut[(b*19 + j)][tap][gamma|beta][C]; rows as wino4_ace_row
hipError_t wino4_style_pack(const float* lut, float* wsty, int B, int C, hipStream_t s);

}  // namespace chk
