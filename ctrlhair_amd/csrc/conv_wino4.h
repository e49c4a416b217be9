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
//   * B operand: lane (n = lane & 15: tile, kk = lane >> 4: channel) transforms ITS 6 x 6 patch (18 LDS reads, 12 one-dimensional
//     transforms of 12 operations) into the 36 B registers of the NEXT k-step while the MFMAs of this one run; the two waves that
//     share the tiles (one per row half) each do it -- the price of the smaller wave tile.
//   * Both operands by LDS-DMA in 16-byte units (buffer_load_dwordx4 ... lds, counted waits): the patch of the tile -- image columns
//     x0 - 4 .. x0 + 35, rows y0 - 1 .. y0 + 32: units are aligned groups of 4 pixels, wholly inside or wholly outside the image (an
//     outside unit's offset lies beyond num_records: zeros) -- and the k-step's 18 KB A image.  Stage = 40 KB, ring of four = all
//     160 KB of LDS, the issue side three k-steps ahead as ONE flat sequence across the block's tasks; one barrier per k-step.
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
constexpr int NST = 4;
constexpr int LDS_BYTES = NST * SUNITS * 16;      // 163 840
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
    int act;                // ACT_* applied to conv + bias + residual
    int reflect;            // 1 = reflection padding (pad 1: row -1 is row 1, row H is row H - 2; columns alike) instead of zeros
    const float* v;         // conv_wino4v.h only: the pre-transformed input (wino4v_pack) -- `in` is then unused
    // set by the launcher
    int nrt, ntx, nty, ntiles, ntasks, nks, rb, tbk;
};

// image of (row tile rt, k-step s): [idx 0..17][lane][4 floats]; float e of idx holds fragment a = 4 idx + e = 36 m + xi (the nine
// reads of a row half are contiguous):  U[xi][row = 32 rt + 16 m + (lane & 15)][ci = 4 s + (lane >> 4)],  U = G g G^T  (xi = 6 i + j)
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
                    for (int i = 0; i < 6; ++i)
                        for (int j = 0; j < 6; ++j) {
                            const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            const int a = m * 36 + i * 6 + j;
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
// one-dimensional output transform  (A^T m):  rows of A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void wino4_out1d(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = __builtin_fmaf(2.f, d2, d1);
    y2 = __builtin_fmaf(4.f, s2, s1);
    y3 = __builtin_fmaf(8.f, d2, d1) + m5;
}

// The two halo columns of a patch row (q[0]: dword 3 of a 16-byte unit, q[5]: dword 0 of the unit after next).  As 4-byte reads every
// lane's address is = 3 (= 0) mod 4 and the 32 lanes of a ds_read_b32 group fall on 8 of its 32 banks (4-way conflict: 53 % of the
// LDS pipe's active cycles in rocprof's SQ_LDS_BANK_CONFLICT, round 5).  Round 6 tried the zero-vector-instruction remedy (-DCH_W4_HALO64):
// aligned 8-byte reads of (q[-1], q[0]) and (q[5], q[6]) -- bank modulus 64, 32 lanes of a group on 32 of 64 banks (2-way), the unused
// halves kept alive by an empty asm so that the compiler does not narrow the reads, the right halo addressed from its own base
// register so that the pair is not fused into one ds_read2_b64 (groups of 16 lanes on 32 banks again).  Measured 2.4 % SLOWER on every
// ResBlock conv shape (profiles/r06_wino4_halo64.txt: 16.64 vs 16.26 ms over the twelve convs of a step, two alternating runs): the
// conflict cycles are not on the critical path of this kernel, the doubled return data is.  The 4-byte reads stay.
typedef float wino4_f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const wino4_f32x2 wino4_lds_f2;
__device__ __forceinline__ wino4_lds_f2* wino4_halo_base(const float* right_halo_row0) {
#ifdef CH_W4_HALO64
    wino4_lds_f2* b = (wino4_lds_f2*)right_halo_row0;
    asm volatile("" : "+v"(b));
    return b;
#else
    return nullptr;
#endif
}
__device__ __forceinline__ void wino4_halo(const float* q, wino4_lds_f2* hb, int r, float& d0, float& d5) {
#ifndef CH_W4_HALO64
    d0 = q[0];
    d5 = q[5];
#else
    const wino4_f32x2 lo = *reinterpret_cast<const wino4_f32x2*>(q - 1), hi = hb[r * (wino4::PUN * 2)];
    asm volatile("" ::"v"(lo.x), "v"(hi.y));
    d0 = lo.y;
    d5 = hi.x;
#endif
}

// Epilogue of one (row tile, spatial tile) task of the plain conv: output transform in registers, bias / residual / activation, 16-byte
// stores.  Lane = tile (tyl, tx) of the 8 x 8 tiles x accumulator rows 32 crt + 16 mh + 4 kk + (0..3).  Shared by wino4_plain_kernel
// and the pre-transformed-input kernel of conv_wino4v.h.
__device__ __forceinline__ void wino4_plain_epilogue(const Wino4Params& p, f32x4 (&acc)[36], int crt, int tile, int mh, int kk, int tyl, int tx) {
    using namespace wino4;
    const int HW = p.H * p.W;
    const int ttx = tile % p.ntx, tty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
    const int y = tty * TS + 4 * tyl, x = ttx * TS + 4 * tx;
    const int rW = p.W >> p.res_up, rHW = rW * (p.H >> p.res_up);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = crt * 32 + mh * 16 + 4 * kk + i, rc = row < p.Cout ? row : p.Cout - 1;
        const float bsv = p.bias ? p.bias[rc] : 0.f;
        f32x4 rr[4];                                 // residual of the (row, tile): loaded first, consumed after the transforms
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
        float t[4][6];                               // A^T M: rows 0..3, columns 0..5
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wino4_out1d(acc[j][i], acc[6 + j][i], acc[12 + j][i], acc[18 + j][i], acc[24 + j][i], acc[30 + j][i], t[0][j], t[1][j], t[2][j], t[3][j]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float o0, o1, o2, o3;
            wino4_out1d(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], o0, o1, o2, o3);
            f32x4 o = {o0 + bsv + rr[r].x, o1 + bsv + rr[r].y, o2 + bsv + rr[r].z, o3 + bsv + rr[r].w};
            if (p.act != ACT_NONE) {
                o.x = apply_act(o.x, p.act); o.y = apply_act(o.y, p.act);
                o.z = apply_act(o.z, p.act); o.w = apply_act(o.w, p.act);
            }
            if (row < p.Cout) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Cout + row) * HW + (y + r) * p.W + x) = o;
        }
        __builtin_amdgcn_sched_barrier(0);           // (one (row, tile) at a time: the accumulators leave little room)
    }
}

// MODE bit 0: reflection padding (the Zencoder's 256 -> 512 conv, architecture.py:174).  Rows: a reflected row is a source offset like any
// other.  Columns: the patch arrives as aligned 4-pixel units, so the unit left of column 0 (right of column W - 1) holds zeros; the one
// element of it a block reads -- its column 0 at the image's left edge, column 5 at the right edge -- is replaced in registers by the
// block's own column 2 / 3 (pixel +1 / W - 2): two selects per patch row, in this instantiation only.
template <int MODE>
__global__ __launch_bounds__(512, 1) void wino4_plain_kernel(const Wino4Params p) {
    using namespace wino4;
    constexpr bool REFL = (MODE & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int mh = wave >> 2, tg = wave & 3;       // row half, tile group
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;
    const int HW = p.H * p.W;
    constexpr unsigned SB = SUNITS * 16u, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

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
    wino_u32x4 d_in, d_a;
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
            int y = y0 + py;
            const int x = x0 + 4 * ux;
            if constexpr (REFL) y = y < 0 ? -y : (y >= p.H ? 2 * p.H - 2 - y : y);
            const bool ok = u < PUNITS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            voff[i] = ok ? (unsigned)(k4 * HW + y * p.W + x) * 4u : 0x80000000u;
        }
        d_in = wino_rsrc(p.in + (long long)ib * p.Cin * HW, (unsigned)p.Cin * HW * 4u);
        d_a = wino_rsrc(p.wpk + (long long)irt * p.nks * ADW, (unsigned)p.nks * ADW * 4u);
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
        so_in += 16u * (unsigned)HW;
        so_a += (unsigned)ADW * 4u;
        if (++is == nk) {
            if (it + G < p.ntasks) {
                it += G;
                is = 0;
                issue_task();
            } else {                   // past the end: keep re-issuing the last k-step (never read; keeps the vmcnt counting uniform)
                is = nk - 1;
                so_in -= 16u * (unsigned)HW;
                so_a -= (unsigned)ADW * 4u;
            }
        }
    };
    auto issue_kstep = [&]() {
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{});
        issue_tail();
    };
    // one k-step's DMAs of this wave (6 / 5 / 4) may still be in flight at the top of a k-step: the two stages it reads were issued before
    auto wait_ring = [&]() {
        if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (wave < 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const int boff = kk * (PPL * 4) + (4 * tyl) * (PUN * 4) + 4 * tx + 3;      // this lane's patch origin (floats) inside a stage
    auto stage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    bool eL = false, eR = false;                  // REFL: this lane's block sits at the left / right image edge (for the patch being transformed)
    auto edge_of = [&](int L, bool& l, bool& r) {
        int rt_, tile_;
        task_of(L < p.ntasks ? L : p.ntasks - 1, rt_, tile_);
        const int ttx_ = tile_ % p.ntx;
        l = ttx_ == 0 && tx == 0;
        r = ttx_ == p.ntx - 1 && tx == 7;
    };
    auto load_row = [&](const float* sp, wino4_lds_f2* hb, int r, float (&d)[6]) {      // patch row r of the lane's tile: 1 + 4 + 1 floats
        const float* q = sp + boff + r * (PUN * 4);
        const f32x4 mid = *reinterpret_cast<const f32x4*>(q + 1);
        d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w;
        wino4_halo(q, hb, r, d[0], d[5]);
        if constexpr (REFL) {
            d[0] = eL ? d[2] : d[0];
            d[5] = eR ? d[3] : d[5];
        }
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(stage(slot) + PSLOTS * 4) + 9 * mh * 64 + lane; };

    issue_kstep();
    issue_kstep();
    issue_kstep();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float v[36], w[36];
    if constexpr (REFL) edge_of(lb, eL, eR);
    {   // B fragments of the first k-step
        const float* sp = stage(lds0);
        wino4_lds_f2* hb = wino4_halo_base(sp + boff + 5);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float d[6];
            load_row(sp, hb, r, d);
            wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], v[6 * r], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wino4_in1d(v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j], v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j]);
    }
    unsigned rslot = lds0;
    // one k-step: nine groups of four MFMAs on the B fragments `vc`; the next k-step's patch is read and transformed into `vx`
    auto kstep = [&](float (&vc)[36], float (&vx)[36]) {
        wait_ring();
        __syncthreads();
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const float* spn = stage(nslot);               // (k-step q + 1 was verified together with q)
        wino4_lds_f2* hbn = wino4_halo_base(spn + boff + 5);
        f32x4 F[2];
        F[0] = ap[0];
        float d[6];
        auto group = [&](auto gt) {
            constexpr int g = decltype(gt)::value;      // xi = 4 g .. 4 g + 3
            if constexpr (g + 1 < 9) F[(g + 1) & 1] = ap[(g + 1) * 64];
            if constexpr (g < 6) load_row(spn, hbn, g, d);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 c = F[g & 1];
            __builtin_amdgcn_s_setprio(1);              // the SIMD's other wave is in its vector section: the matrix pipe goes first (-1.5 % measured)
            acc[4 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, vc[4 * g], acc[4 * g], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, vc[4 * g + 1], acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, vc[4 * g + 2], acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, vc[4 * g + 3], acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (g < 6)                        // row transform of patch row g (behind the MFMAs: its LDS reads land meanwhile)
                wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], vx[6 * g], vx[6 * g + 1], vx[6 * g + 2], vx[6 * g + 3], vx[6 * g + 4], vx[6 * g + 5]);
            if constexpr (g >= 6) {                     // column transforms 2 (g - 6), 2 (g - 6) + 1, in place
#pragma unroll
                for (int j = 2 * (g - 6); j < 2 * (g - 6) + 2; ++j)
                    wino4_in1d(vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j], vx[30 + j], vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j],
                               vx[30 + j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g >= 2 && g < 6) issue_piece(WInt<g - 2>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
        group(WInt<6>{}); group(WInt<7>{}); group(WInt<8>{});
        issue_tail();
        rslot = nslot;
    };

    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        for (int cs = 0; cs < nk; cs += 2) {
            kstep(v, w);          // (nks is even: the launcher)
            if constexpr (REFL)
                if (cs + 2 >= nk) edge_of(ct + G, eL, eR);      // the task's last k-step transforms the NEXT task's first patch
            kstep(w, v);
        }
        // ---- epilogue of task ct ---------------------------------------------------------------------------------------------
        int crt, tile;
        task_of(ct, crt, tile);
        wino4_plain_epilogue(p, acc, crt, tile, mh, kk, tyl, tx);
#pragma unroll
        for (int x2 = 0; x2 < 36; ++x2) acc[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's loads / stores share the counter with the ring: drain once per task
    }
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
//   * the B fragments of a task's first k-step are recomputed from the staged patch after the epilogue instead of being carried
//     through it (the modulation needs the registers).
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
    const float* v;         // conv_wino4v.h only: the pre-transformed hidden activations (wino4v_pack) -- `actv` is then unused
    int nrt, ntx, nty, ntiles, ntasks, nks, rb, tbk;      // set by the launcher
};
// GEMM row R of the packed SPADE image -> (channel, beta)
__host__ __device__ inline void wino4_ace_row(int R, int& ch, int& beta) {
    const int rt = R >> 5, m = (R >> 4) & 1, r = R & 15;
    ch = rt * 16 + m * 8 + (r >> 2) * 2 + (r & 1);
    beta = (r >> 1) & 1;
}

// ACE epilogue of one (row tile, spatial tile) task: output transform of gamma (accumulator rows 0, 1) and beta (rows 2, 3) of the
// lane's two channels 16 crt + 8 mh + 2 kk + (0, 1), noise + eval-BN + modulation + activation, 16-byte stores.  Shared by
// wino4_ace_kernel and the pre-transformed-input kernel of conv_wino4v.h.
__device__ __forceinline__ void wino4_ace_epilogue(const Wino4AceParams& p, f32x4 (&acc)[36], int crt, int tile, int mh, int kk, int tyl, int tx) {
    using namespace wino4;
    const int HW = p.H * p.W;
    const int ttx = tile % p.ntx, tty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
    const int y = tty * TS + 4 * tyl, x = ttx * TS + 4 * tx;
    const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
    f32x4 nz[4];                                  // nz[c] = noise of column x + c, rows y .. y + 3 (plane layout [W][H])
    const float* nzp = p.noise + (long long)b * p.noise_bstride + (long long)x * p.H + y;
#pragma unroll
    for (int c = 0; c < 4; ++c) nz[c] = *reinterpret_cast<const f32x4*>(nzp + (long long)c * p.H);
    const int cA = crt * 16 + mh * 8 + 2 * kk;
#pragma unroll
    for (int e = 0; e < 2; ++e) {                 // the lane's two channels: accumulator rows e (gamma) and 2 + e (beta)
        const int ch = cA + e, cc = ch < p.C ? ch : p.C - 1;
        const float gb = 1.f + p.bias_g[cc], bb = p.bias_b[cc], pa = p.bn_a[cc], pd = p.bn_d[cc], pn = p.nv[cc];
        f32x4 xr[4];                              // x rows y .. y + 3, columns x .. x + 3
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
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            wino4_out1d(acc[j][e], acc[6 + j][e], acc[12 + j][e], acc[18 + j][e], acc[24 + j][e], acc[30 + j][e], tg_[0][j], tg_[1][j], tg_[2][j],
                        tg_[3][j]);
            wino4_out1d(acc[j][2 + e], acc[6 + j][2 + e], acc[12 + j][2 + e], acc[18 + j][2 + e], acc[24 + j][2 + e], acc[30 + j][2 + e], tb_[0][j],
                        tb_[1][j], tb_[2][j], tb_[3][j]);
        }
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

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino4_ace_kernel(const Wino4AceParams p) {
    using namespace wino4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int mh = wave >> 2, tg = wave & 3;
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;                          // 32, or 38 with the style images
    const int AP = wino_apitch(p.W), APL = p.H * AP;
    constexpr unsigned SB = SUNITS * 16u, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;
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

    // ---- issue side (as wino4_plain_kernel; rows outside the image: offset beyond num_records; columns: the planes' zero pads) ----
    unsigned voff[3];
    const unsigned va = (unsigned)tid * 16u;
    int it = lb, is = 0;
    wino_u32x4 d_in, d_a, d_s;
    unsigned so_in = 0, so_a = 0;
    auto issue_task = [&]() {
        int irt, tile;
        task_of(it, irt, tile);
        const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, ib = tile / (p.ntx * p.nty);
        const int y0 = ty * TS - 1, x0 = tx * TS - 4 + WINO_AXOFF;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int u = i * 512 + tid;
            const int k4 = u / PPL, rem = u - k4 * PPL;
            const int py = rem / PUN, ux = rem - py * PUN;
            const int y = y0 + py;
            const bool ok = u < PUNITS && (unsigned)y < (unsigned)p.H;
            voff[i] = ok ? (unsigned)(k4 * APL + y * AP + x0 + 4 * ux) * 4u : 0x80000000u;
        }
        const int K = 128 + (p.wsty ? 20 : 0);
        d_in = wino_rsrc(p.actv + (long long)ib * K * APL, (unsigned)K * APL * 4u);
        d_a = wino_rsrc(p.wpk + (long long)irt * 32 * ADW, 32u * (unsigned)ADW * 4u);
        if (p.wsty) d_s = wino_rsrc(p.wsty + ((long long)ib * p.nrt + irt) * 6 * ADW, 6u * (unsigned)ADW * 4u);
        so_in = 0;
        so_a = 0;
    };
    issue_task();
    unsigned islot = lds0;
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
        so_in += 16u * (unsigned)APL;
        so_a += (unsigned)ADW * 4u;
        ++is;
        if (is == 32 && nk > 32) {         // the style images of the task's sample follow the hidden channels
            d_a = d_s;
            so_a = 0;
        }
        if (is == nk) {
            if (it + G < p.ntasks) {
                it += G;
                is = 0;
                issue_task();
            } else {                   // past the end: keep re-issuing the last k-step (never read; keeps the vmcnt counting uniform)
                is = nk - 1;
                so_in -= 16u * (unsigned)APL;
                so_a -= (unsigned)ADW * 4u;
            }
        }
    };
    auto issue_kstep = [&]() {
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{});
        issue_tail();
    };
    auto wait_ring = [&]() {
        if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (wave < 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const int boff = kk * (PPL * 4) + (4 * tyl) * (PUN * 4) + 4 * tx + 3;
    auto stage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    auto load_row = [&](const float* sp, wino4_lds_f2* hb, int r, float (&d)[6]) {
        const float* q = sp + boff + r * (PUN * 4);
        const f32x4 mid = *reinterpret_cast<const f32x4*>(q + 1);
        d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w;
        wino4_halo(q, hb, r, d[0], d[5]);
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(stage(slot) + PSLOTS * 4) + 9 * mh * 64 + lane; };
    auto first_v = [&](unsigned slot, float (&vv)[36]) {          // B fragments of the k-step staged in `slot`
        const float* sp = stage(slot);
        wino4_lds_f2* hb = wino4_halo_base(sp + boff + 5);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float d[6];
            load_row(sp, hb, r, d);
            wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], vv[6 * r], vv[6 * r + 1], vv[6 * r + 2], vv[6 * r + 3], vv[6 * r + 4], vv[6 * r + 5]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wino4_in1d(vv[j], vv[6 + j], vv[12 + j], vv[18 + j], vv[24 + j], vv[30 + j], vv[j], vv[6 + j], vv[12 + j], vv[18 + j], vv[24 + j], vv[30 + j]);
    };

    issue_kstep();
    issue_kstep();
    issue_kstep();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned rslot = lds0;
    auto kstep = [&](float (&vc)[36], float (&vx)[36]) {
        wait_ring();
        __syncthreads();
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const float* spn = stage(nslot);
        wino4_lds_f2* hbn = wino4_halo_base(spn + boff + 5);
        f32x4 F[2];
        F[0] = ap[0];
        float d[6];
        auto group = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            if constexpr (g + 1 < 9) F[(g + 1) & 1] = ap[(g + 1) * 64];
            if constexpr (g < 6) load_row(spn, hbn, g, d);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 c = F[g & 1];
            __builtin_amdgcn_s_setprio(1);
            acc[4 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, vc[4 * g], acc[4 * g], 0, 0, 0);
            acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, vc[4 * g + 1], acc[4 * g + 1], 0, 0, 0);
            acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, vc[4 * g + 2], acc[4 * g + 2], 0, 0, 0);
            acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, vc[4 * g + 3], acc[4 * g + 3], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (g < 6)
                wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], vx[6 * g], vx[6 * g + 1], vx[6 * g + 2], vx[6 * g + 3], vx[6 * g + 4], vx[6 * g + 5]);
            if constexpr (g >= 6) {
#pragma unroll
                for (int j = 2 * (g - 6); j < 2 * (g - 6) + 2; ++j)
                    wino4_in1d(vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j], vx[30 + j], vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j],
                               vx[30 + j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g >= 2 && g < 6) issue_piece(WInt<g - 2>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
        group(WInt<6>{}); group(WInt<7>{}); group(WInt<8>{});
        issue_tail();
        rslot = nslot;
    };

    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        {
            float v[36], w[36];
            first_v(rslot, v);            // (the stage was verified by the last barrier: every k-step's barrier covers two stages)
            for (int cs = 0; cs < nk; cs += 2) {
                kstep(v, w);              // (nk is even: 32 or 38)
                kstep(w, v);
            }
        }
        // ---- ACE epilogue of task ct: this lane = tile (tyl, tx) x channels cA, cA + 1 ---------------------------------------------
        int crt, tile;
        task_of(ct, crt, tile);
        wino4_ace_epilogue(p, acc, crt, tile, mh, kk, tyl, tx);
#pragma unroll
        for (int x2 = 0; x2 < 36; ++x2) acc[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

inline bool wino4_supported(int H, int W, int Cin) { return H % wino4::TS == 0 && W % wino4::TS == 0 && Cin % 8 == 0 && Cin >= 16; }
// Does F(4x4,3x3) beat F(2x2,3x3) on this launch?  Both kernels run one task per CU at a time; an F(4x4) task (32 x 32 pixels x 32 rows) is
// 36 MFMAs per wave and k-step at ~58 % of the matrix pipe, the two F(2x2) tasks covering the same pixels (32 x 16 each) are 32 each at
// ~68 %.  With many tasks per CU F(4x4) wins 36 : 64; with fewer tasks than CUs both take one round and the smaller F(2x2) task is the
// shorter one (batch-1 rendering: 4.54 -> 4.21 ms at 256^2, 6.34 -> 5.92 ms at 512^2 with F(2x2) everywhere, tools/lat_b1.py).
inline bool wino4_pays(long long ntasks4, int cus) {
    if (cus <= 0) cus = 256;
    const long long r4 = (ntasks4 + cus - 1) / cus, r2 = (2 * ntasks4 + cus - 1) / cus;
    return r4 * 36 * 68 < r2 * 32 * 58;
}
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
