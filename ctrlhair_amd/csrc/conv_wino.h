// conv_wino.h -- 3x3 stride-1 zero-padded convolutions as Winograd F(2x2, 3x3) on the exact-f32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32): 16 multiplies per 2x2 output quad and channel instead of 36, every product and sum an IEEE
// f32 operation (the transforms are f32 adds, the contraction is the MFMA's f32 fma chain).  A real-arithmetic identity of
// the reference's conv2d (architecture.py:82-91 conv_0 / conv_1; normalization.py:249-257 mlp_gamma / mlp_beta), like the
// other reformulations of DESIGN.md section 2; measured deviation from the direct evaluation <= 4e-6 on the generator output
// (tests/test_hip_wino.py, tests/test_winograd_model.py).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        g: 3x3 kernel, d: 4x4 input patch, Y: 2x2 output quad
//   M[xi][row][quad] = sum_ci U[xi][row][ci] * V[xi][ci][quad]            (xi = 0..15: sixteen independent GEMMs)
//
// Mapping to the hardware
//   * One persistent 512-thread block per CU (grid = #CUs), 8 waves = 2 per SIMD.  A block task = a spatial tile of 32 x 16
//     pixels (16 x 8 quads) of one sample x a ROW TILE of 32 GEMM rows; wave w owns quad row w (16 quads) for all 32 rows and
//     all 16 xi: 32 accumulators of 16x16 (128 registers).  One k-step = 4 input channels = 32 MFMAs per wave.
//   * B operand: lane (n = lane & 15, kk = lane >> 4) holds quad n, channel kk of the k-step -- exactly one (quad, channel)
//     pair per lane, so the lane transforms ITS OWN 4x4 patch (8 ds_read_b64 from the un-expanded LDS patch + 32 f32 adds)
//     and the 16 results are the B registers of the 16 xi.  V never exists in memory.
//   * A operand: U = G g G^T computed in double at ch_finalize, rounded once to f32 and packed into per-lane fragment order
//     (pack_wino_A); a k-step's 8 KB image goes L2 -> LDS by LDS-DMA (global_load_lds, 16 B per lane) and is shared by the 8
//     waves (ds_read_b128, four fragments per read).
//   * Input patch (tile + halo, 4 channels): LDS-DMA too, 4 B per lane with per-lane source addresses; taps outside the
//     image read a zero word.  No vector register ever holds staged data, so the MFMA stream waits on nothing but the
//     counted s_waitcnt in front of the one barrier per k-step.
//   * A 6-stage ring of k-steps runs as ONE flat sequence across the block's tasks: the next task's first k-steps are in
//     flight during the current task's last ones and its epilogue (no exposed prologue).  The first fragments of k-step
//     q + 1 are read before the barrier that ends k-step q (its data were verified one barrier earlier), so the MFMA stream
//     continues across barriers.
//   * The ResBlock's learned 1x1 shortcut conv_s (architecture.py:75-79) stays a direct 1x1 GEMM (conv_mfma.h) whose output this
//     kernel adds as the residual: folded into the Winograd accumulators it costs the same MFMAs (a centre-tap kernel is
//     non-zero at the four central xi) but 8 MFMAs per barrier interval do not cover the interval's fixed costs (measured:
//     +1.7 ms per fused layer against +0.8 for the separate launch, tools/wino_bench.hip at an earlier revision).
//   * Epilogue: output transform in registers (24 adds per (row, quad)), bias / residual / activation, float2 stores
//     (16 lanes = one 128-byte line).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "conv_mfma.h"

namespace chk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void wino_lds_void;
typedef __attribute__((address_space(1))) const void wino_glb_void;

namespace wino {
constexpr int TW = 32, TH = 16;              // spatial tile (pixels)
constexpr int PWP = 36, PROWS = TH + 2;      // LDS patch: 18 rows of 36 floats (34 used)
constexpr int PS = 672;                      // plane stride (floats): 648 used; 672 = 10 * 64 + 32 -> the two channel planes a
                                             //   32-lane ds_read_b64 group touches sit on disjoint bank halves
constexpr int NPD = 6;                       // patch DMA instructions per thread per k-step (6 * 512 >= 4 * 672)
constexpr int PDW = NPD * 512;               // patch dwords per stage
constexpr int ADW = 2048;                    // A dwords per stage (32 fragments x 64 lanes)
constexpr int SDW = PDW + ADW;               // 20 KB per stage
constexpr int NST = 6;                       // ring depth
constexpr int NLD = NPD + 1;                 // vector-memory instructions per thread per k-step
constexpr int LDS_BYTES = NST * SDW * 4;     // 120 KB
}  // namespace wino

struct WinoParams {
    const float* in;        // [B][Cin][H][W]
    const float* wpk;       // pack_wino_A image
    float* out;             // [B][Cout][H][W]
    int B, Cin, Cout, H, W; // H % 16 == 0, W % 32 == 0, Cin % 4 == 0
    const float* bias;      // [Cout] or null
    const float* res;       // [B][Cout][H >> res_up][W >> res_up] or null
    int res_up, act;
    int reflect;            // 1 = reflection padding (pad 1: row -1 is row 1, row H is row H - 2) instead of zeros
    int in_up;              // 1 = the input is [B][Cin][H / 2][W / 2], read through its nearest x2 view (generator.py:53-style up-sampling folded
                            //   into the conv: a per-lane source offset like the padding)
    int d2s;                // 1 = depth-to-space store: GEMM row 4 c + (2 py + px) holds phase (py, px) of channel c, out is [B][Cout / 4][2 H][2 W]
                            //   (a stride-2 ConvTranspose2d as four phase convs of the input grid; no residual)
    int pair16;             // set by the launcher for 16 x 16 images (B even): two samples side by side fill a tile of 32 x 16 pixels
    const float* zero;      // >= 16 bytes of zeros in device memory (source of out-of-image patch elements)
    unsigned* claim;        // eight zeroed counters: dynamic task claiming (below; Cin >= 24), or null: static split
    float* partial;         // split-K workspace of partial_cap floats, or null.  Launches with far fewer tasks than CUs (single images on the deep
    long long partial_cap;  //   levels: 64 tasks of 256 k-steps each) are split over the input channels: `ksplit` slices, each a task of its own
                            //   that writes its raw output-transformed sums to partial[slice][B][Cout][H][W]; wino_splitk_reduce adds the
                            //   slices in order (deterministic) and applies bias / residual / activation
    // set by the launcher
    int nrt, ntx, nty, ntiles, ntasks, nks, rb, tbk;
    int ksplit;             // 1, or the number of K slices (ntasks counts every (slice, task))
};

// ---- host: weight transform + packing -------------------------------------------------------------------------------------
// image of (row tile rt, k-step s): [idx 0..7][lane][4 floats]; float e of idx holds fragment a = 4 idx + e = (xi = a >> 1,
// m = a & 1): U[xi][row = 32 rt + 16 m + (lane & 15)][ci = 4 s + (lane >> 4)]
template <class F>
std::vector<float> pack_wino_A(int rows, int Cin, F get) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nrt = (rows + 31) / 32, nks = Cin / 4;
    std::vector<float> dst((size_t)nrt * nks * 2048, 0.f);
    for (int rt = 0; rt < nrt; ++rt)
        for (int s = 0; s < nks; ++s) {
            float* img = dst.data() + ((size_t)rt * nks + s) * 2048;
            for (int m = 0; m < 2; ++m)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = rt * 32 + m * 16 + (lane & 15), ci = 4 * s + (lane >> 4);
                    if (row >= rows) continue;
                    double g[3][3], t[4][3];
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) g[a][b] = get(row, ci, a * 3 + b);
                    for (int i = 0; i < 4; ++i)
                        for (int b = 0; b < 3; ++b) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
                    for (int i = 0; i < 4; ++i)
                        for (int j = 0; j < 4; ++j) {
                            const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                            const int a = (i * 4 + j) * 2 + m;
                            img[((a >> 2) * 64 + lane) * 4 + (a & 3)] = (float)u;
                        }
                }
        }
    return dst;
}
// ---- device ------------------------------------------------------------------------------------------------------------------
// task L -> (row tile, spatial tile).  32 consecutive tasks (what the 32 CUs of one XCD run at the same time) form a block of
// `rb` row tiles x `tbk` spatial tiles, so that their A images and input patches are shared through the XCD's L2.
__device__ __forceinline__ void wino_task(const WinoParams& p, int L, int& rt, int& tile) {
    const int per = p.tbk * p.nrt;
    const int tg = L / per;
    int r = L - tg * per;
    const int tgsz = min(p.tbk, p.ntiles - tg * p.tbk);
    const int rg = r / (tgsz * p.rb);
    r -= rg * tgsz * p.rb;
    const int rgsz = min(p.rb, p.nrt - rg * p.rb);
    const int tl = r / rgsz;
    rt = rg * p.rb + (r - tl * rgsz);
    tile = tg * p.tbk + tl;
}

__device__ __forceinline__ void wino_in_transform(const float (&d)[4][4], float (&v)[16]) {
    float t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[4 * i + 0] = t[i][0] - t[i][2];
        v[4 * i + 1] = t[i][1] + t[i][2];
        v[4 * i + 2] = t[i][2] - t[i][1];
        v[4 * i + 3] = t[i][1] - t[i][3];
    }
}

__device__ __forceinline__ void wino_out_transform(const float (&M)[16], float& y00, float& y01, float& y10, float& y11) {
    float s0[4], s1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s0[r] = M[4 * r] + M[4 * r + 1] + M[4 * r + 2];
        s1[r] = M[4 * r + 1] - M[4 * r + 2] - M[4 * r + 3];
    }
    y00 = s0[0] + s0[1] + s0[2];
    y01 = s1[0] + s1[1] + s1[2];
    y10 = s0[1] - s0[2] - s0[3];
    y11 = s1[1] - s1[2] - s1[3];
}

typedef unsigned wino_u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptor (gfx9 family): base, stride 0, num_records bytes, 32-bit raw format word
__device__ __forceinline__ wino_u32x4 wino_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    wino_u32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xFFFFu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}
// LDS-DMA through inline asm: hipcc orders every later ds_read behind a builtin LDS-DMA with s_waitcnt vmcnt(0) (it cannot prove
// the ring slots disjoint), which drained the ring once per k-step; an asm statement is invisible to that bookkeeping, the
// counted waits are ours.  buffer_load ... lds: per-lane 32-bit byte offsets (an offset >= num_records returns 0: the zero
// padding of the conv costs nothing), the k-step's channel offset in an SGPR; M0 = LDS destination of lane 0.
__device__ __forceinline__ void wino_dma4(unsigned voff, const wino_u32x4& d, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds" ::"v"(voff), "s"(d), "s"(soff), "s"(lds) : "memory");
}
__device__ __forceinline__ void wino_dma16(unsigned voff, const wino_u32x4& d, unsigned soff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(d), "s"(soff), "s"(lds) : "memory");
}

// ---- dynamic task claiming (round 5) -----------------------------------------------------------------------------------------------
// A persistent kernel that splits its tasks statically (block b takes tasks b, b + G, ...) ends when its LAST block does: a block
// that starts late -- because a kernel of another stream (label tables, interior passes: HBM-bound work that runs beside these
// matrix-bound convs on a few CUs, sean_model.cpp) held its CU -- holds the whole launch back by as much.  With `claim` set the
// blocks take their tasks from eight counters, one per XCD (device memory, zeroed before the launch): claim number j of XCD x is
// task ((j >> cs) * 8 + x) << cs | (j & (2^cs - 1)) -- chunks of 2^cs consecutive tasks stay on one XCD, as the static order's
// xcd_remap arranges (their A images / patches meet in that XCD's L2), and the XCDs' shares differ by at most one chunk.  A block
// claims its first two tasks when it starts and then, before the epilogue of task k, the task after next; the answer is read
// after that epilogue's own `s_waitcnt vmcnt(0)` and handed to the other waves through LDS (one more barrier per task).  Which
// block computes a task never changes what is computed: results stay bit-identical from run to run.
__device__ __forceinline__ int wino_xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (2 << 11)) & 7u); }      // HW_REG_XCC_ID[2:0]
__device__ __forceinline__ int wino_claim_task(unsigned j, int xcc, int cs, int ntasks) {
    const long long t = ((((long long)(j >> cs) << 3) + xcc) << cs) + (j & ((1u << cs) - 1u));
    return t < ntasks ? (int)t : ntasks;
}
// a word another wave of this CU stored a few microseconds ago, read past the scalar cache (SGPR result: lgkmcnt, not the ring's vmcnt)
__device__ __forceinline__ int wino_mail_read(const unsigned* p) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return (int)v;
}
// chunk size of a launch with `ntasks` tasks (log2): 32 where every XCD gets several chunks, smaller for short launches
__host__ __device__ inline int wino_claim_cs(int ntasks) { return ntasks >= 2048 ? 5 : (ntasks >= 1024 ? 4 : (ntasks >= 512 ? 3 : (ntasks >= 256 ? 2 : 0))); }

#ifndef WINO_NG
#define WINO_NG 8      // MFMA groups per k-step between scheduling fences: 8 (of four MFMAs) or 4 (of eight)
#endif
template <int N>
struct WInt { static constexpr int value = N; };

// MODE bit 0: the depth-to-space epilogue (WinoParams::d2s); bit 1: nearest x2 input view (WinoParams::in_up) -- their own
// instantiations, so that the generator's main one is untouched
template <int MODE>
__global__ __launch_bounds__(512, 1) void wino_plain_kernel(const WinoParams p) {
    using namespace wino;
    constexpr int D2S = MODE & 1;
    constexpr bool UP = (MODE & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int G = gridDim.x;
    // static split: block lb takes tasks lb, lb + G, ...; dynamic (p.claim set; the launcher guarantees nks >= NST, so that the issue
    // side is never more than one task ahead of the consumers): tasks claimed from the XCD's counter, see wino_claim_task
    __shared__ int mbox[2];
    const bool dyn = p.claim != nullptr && p.ntasks >= 8 * G;      // (a block holds two claimed tasks: only where a block gets many)
    int lb, tn;                                    // first task, the task after it (ntasks: none)
    if (dyn) {
        if (tid == 0) {
            const int xcc = wino_xcc_id(), ccs = wino_claim_cs(p.ntasks);
            const unsigned j0 = __hip_atomic_fetch_add(p.claim + xcc, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mbox[0] = wino_claim_task(j0, xcc, ccs, p.ntasks);
            mbox[1] = wino_claim_task(j0 + 1u, xcc, ccs, p.ntasks);
        }
        __syncthreads();
        lb = __builtin_amdgcn_readfirstlane(mbox[0]);
        tn = __builtin_amdgcn_readfirstlane(mbox[1]);
        __syncthreads();
    } else {
        lb = xcd_remap(blockIdx.x, G);
        tn = p.ntasks;
    }
    if (lb >= p.ntasks) return;
    int ct = lb;                                   // the task the consumers work on
    const int KSP = (D2S == 0 && p.ksplit > 1) ? p.ksplit : 1;            // K slices: task index = slice * nbase + (row tile, tile) index
    const int nbase = p.ntasks / KSP;
    const int nk = p.nks / KSP;
    const int HW = p.H * p.W;
    const int Ws = UP ? p.W >> 1 : p.W, HWs = UP ? HW >> 2 : HW;          // the input planes as stored
    constexpr unsigned SB = SDW * 4, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;          // LDS byte address of the ring

    // ---- issue side (everything wave-uniform except the six patch offsets) -----------------------------------------------------
    unsigned voff[NPD];
    const unsigned va = (unsigned)tid * 16u;
    int it = lb, is = 0;                   // task / k-step the next issue serves
    wino_u32x4 d_in, d_a;
    unsigned so_in = 0, so_a = 0;
    auto issue_task = [&]() {
        int irt, tile;
        const int isp = it / nbase;            // K slice of the task
        wino_task(p, it - isp * nbase, irt, tile);
        const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, ib = tile / (p.ntx * p.nty);
        const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
        for (int i = 0; i < NPD; ++i) {        // element e of the stage's patch image -> (channel k4, patch row, patch column)
            const int e = i * 512 + tid;
            const int k4 = e / PS, rem = e - k4 * PS;
            const int py = rem / PWP, px = rem - py * PWP;
            int y = y0 + py, x = x0 + px;
            int sub = 0;                       // pair16: patch columns 0-17 = sample 2 ib, 18-35 = sample 2 ib + 1 (each 16 wide + its halo)
            if (p.pair16) {
                sub = px >= 18 ? 1 : 0;
                x = px - 18 * sub - 1;
            }
            if (p.reflect) {                   // (architecture.py:159 ReflectionPad2d(1): a per-lane source offset like any other)
                y = y < 0 ? -y : (y >= p.H ? 2 * p.H - 2 - y : y);
                x = x < 0 ? -x : (x >= p.W ? 2 * p.W - 2 - x : x);
            }
            const bool ok = k4 < 4 && rem < PROWS * PWP && (p.pair16 || px < TW + 2) && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            voff[i] = ok ? (unsigned)((sub * p.Cin + k4) * HWs + (UP ? (y >> 1) * Ws + (x >> 1) : y * p.W + x)) * 4u : 0x80000000u;
        }
        const long long c0 = (long long)isp * nk * 4;                       // first input channel of the slice
        d_in = p.pair16 ? wino_rsrc(p.in + ((long long)ib * 2 * p.Cin + c0) * HWs, (unsigned)((2 * ib + 1 < p.B ? 2 : 1) * p.Cin - c0) * HWs * 4u)
                        : wino_rsrc(p.in + ((long long)ib * p.Cin + c0) * HWs, (unsigned)(p.Cin - c0) * HWs * 4u);
        d_a = wino_rsrc(p.wpk + ((long long)irt * p.nks + (long long)isp * nk) * 2048, (unsigned)nk * 8192u);
        so_in = 0;
        so_a = 0;
    };
    issue_task();
    unsigned islot = lds0;                 // ring slot the next issue fills
    // issue_part(part): part 0..2 = two patch DMAs each, part 3 = the A image + advance.  The parts sit between the MFMA groups of
    // a k-step, so that they issue in the shadow of the matrix pipe.
    auto issue_part = [&](int part) {
        const unsigned wb = islot + (unsigned)wave * 256u;
        if (part < 3) {
#pragma unroll
            for (int i = 2 * part; i < 2 * part + 2; ++i) {
                wino_dma4(voff[i], d_in, so_in, wb + (unsigned)i * 2048u);
            }
            return;
        }
        wino_dma16(va, d_a, so_a, islot + PDW * 4u + (unsigned)wave * 1024u);
        islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
        so_in += 16u * (unsigned)HWs;
        so_a += 8192u;
        if (++is == nk) {
            const int nx = dyn ? (it == ct ? tn : p.ntasks) : it + G;
            if (nx < p.ntasks) {
                it = nx;
                is = 0;
                issue_task();
            } else {                       // past the end: keep re-issuing the last k-step (never read; keeps the vmcnt counting uniform)
                is = nk - 1;
                so_in -= 16u * (unsigned)HWs;
                so_a -= 8192u;
            }
        }
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[x][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int boff = kk * PS + (2 * wave) * PWP + 2 * n + ((p.pair16 && n >= 8) ? 2 : 0);      // this lane's patch origin (floats) inside a stage
    auto load_raw = [&](unsigned slot, float (&d)[4][4]) {
        const float* sp = reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + boff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 lo = *reinterpret_cast<const float2*>(sp + i * PWP), hi = *reinterpret_cast<const float2*>(sp + i * PWP + 2);
            d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
        }
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + PDW) + lane; };

#pragma unroll
    for (int j = 0; j < NST - 1; ++j) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(part);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (NST - 3)) : "memory");
    __syncthreads();
    float v[16];
    f32x4 a0, a1;
    {
        float d[4][4];
        load_raw(lds0, d);
        wino_in_transform(d, v);
        const f32x4* ap = a_ptr(lds0);
        a0 = ap[0];
        a1 = ap[64];
    }

    unsigned rslot = lds0;
    bool after_epi = false;
    float w[16];                                                   // the other half of the (v, w) ping-pong of B fragments
    // one k-step: MFMAs on the B fragments `vc` (ready) while the next k-step's patch is read and transformed into `vx`
    auto kstep = [&](float (&vc)[16], float (&vx)[16]) {
        if (after_epi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (NST - 3)) : "memory");
        __syncthreads();
        after_epi = false;
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const f32x4* apn = a_ptr(nslot);
        float dn[4][4], t[4][4];
        {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const f32x4 c0 = a0, c1 = a1;
                if (h < 3) {
                    a0 = ap[(2 * h + 2) * 64];
                    a1 = ap[(2 * h + 3) * 64];
                } else {
                    a0 = apn[0];
                    a1 = apn[64];
                }
                if (h == 0) load_raw(nslot, dn);                   // k-step q + 1 was verified by the barrier above
                __builtin_amdgcn_sched_barrier(0);                 // the LDS reads of the NEXT group first: this group's MFMA time
                                                                   //   is their latency
                // the next k-step's input transform, spread over the MFMA groups (B^T d, then (.) B two rows at a time)
                if (h == 1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        t[0][j] = dn[0][j] - dn[2][j];
                        t[1][j] = dn[1][j] + dn[2][j];
                        t[2][j] = dn[2][j] - dn[1][j];
                        t[3][j] = dn[1][j] - dn[3][j];
                    }
                }
                if (h >= 2) {
#pragma unroll
                    for (int i = 2 * (h - 2); i < 2 * (h - 2) + 2; ++i) {
                        vx[4 * i + 0] = t[i][0] - t[i][2];
                        vx[4 * i + 1] = t[i][1] + t[i][2];
                        vx[4 * i + 2] = t[i][2] - t[i][1];
                        vx[4 * i + 3] = t[i][1] - t[i][3];
                    }
                }
                {
                    acc[4 * h][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, vc[4 * h], acc[4 * h][0], 0, 0, 0);
                    acc[4 * h][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, vc[4 * h], acc[4 * h][1], 0, 0, 0);
                    acc[4 * h + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.z, vc[4 * h + 1], acc[4 * h + 1][0], 0, 0, 0);
                    acc[4 * h + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.w, vc[4 * h + 1], acc[4 * h + 1][1], 0, 0, 0);
                    acc[4 * h + 2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, vc[4 * h + 2], acc[4 * h + 2][0], 0, 0, 0);
                    acc[4 * h + 2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, vc[4 * h + 2], acc[4 * h + 2][1], 0, 0, 0);
                    acc[4 * h + 3][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.z, vc[4 * h + 3], acc[4 * h + 3][0], 0, 0, 0);
                    acc[4 * h + 3][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.w, vc[4 * h + 3], acc[4 * h + 3][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                issue_part(h);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        rslot = nslot;
    };
    for (int k = 0;; ++k) {
        for (int cs = 0; cs < nk; cs += 2) {
            kstep(v, w);          // (nks is even: wino_supported)
            kstep(w, v);
        }
        const bool more = dyn ? tn < p.ntasks : ct + G < p.ntasks;
        unsigned jnext = 0;       // dynamic claiming: the task after next, answered during the epilogue
        if (dyn && more && tid == 0) jnext = __hip_atomic_fetch_add(p.claim + wino_xcc_id(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- epilogue of task ct: every load first (rows clamped, so no load sits behind a branch: the per-row load -> use chains
        //      of the first version cost four memory round trips per task), then the output transforms, then the stores -----------
        if constexpr (D2S != 0) {
            // depth-to-space store (WinoParams::d2s): a lane's four rows of a 16-row half are the four phases of ONE channel, so its
            // quad becomes a 4 x 4 block of the output at twice the resolution: four 16-byte stores per channel
            int crt, tile;
            wino_task(p, ct, crt, tile);
            const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
            const int y = ty * TH + 2 * wave, x = tx * TW + 2 * n;
            const int Cr = p.Cout >> 2, W2 = 2 * p.W;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int c = crt * 8 + m * 4 + kk;
                const float bsv = p.bias ? p.bias[c < Cr ? c : Cr - 1] : 0.f;
                float yv[4][4];                                      // [phase][quad pixel 2 a + b]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float M[16];
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][m][i];
                    wino_out_transform(M, yv[i][0], yv[i][1], yv[i][2], yv[i][3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        yv[i][e] += bsv;
                        if (p.act != ACT_NONE) yv[i][e] = apply_act(yv[i][e], p.act);
                    }
                }
                if (c < Cr) {
                    float* op = p.out + (((long long)b * Cr + c) * 2 * p.H + 2 * y) * W2 + 2 * x;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int py = 0; py < 2; ++py)       // output row 2 (y + a) + py: columns 2 x .. 2 x + 3 = (b, px) = (0,0) (0,1) (1,0) (1,1)
                            *reinterpret_cast<float4*>(op + (long long)(2 * a + py) * W2) =
                                make_float4(yv[2 * py][2 * a], yv[2 * py + 1][2 * a], yv[2 * py][2 * a + 1], yv[2 * py + 1][2 * a + 1]);
                }
            }
#pragma unroll
            for (int x2 = 0; x2 < 16; ++x2)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[x2][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            after_epi = true;
        } else {
            int crt, tile;
            const int csp = ct / nbase;
            wino_task(p, ct - csp * nbase, crt, tile);
            const bool slice = KSP > 1;                  // a K slice: raw sums into its slab (bias / residual / activation: wino_splitk_reduce)
            float* const obase = slice ? p.partial + (long long)csp * p.B * p.Cout * HW : p.out;
            const float* const rbase = slice ? nullptr : p.res;
            const float* const bbase = slice ? nullptr : p.bias;
            const int eact = slice ? ACT_NONE : p.act;
            const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty;
            const int b0 = p.pair16 ? 2 * (tile / (p.ntx * p.nty)) + (n >> 3) : tile / (p.ntx * p.nty);
            const bool bval = b0 < p.B;                  // (pair16 with an odd batch: the last tile's second sample does not exist)
            const int b = bval ? b0 : p.B - 1;
            const int y = ty * TH + 2 * wave, x = p.pair16 ? 2 * (n & 7) : tx * TW + 2 * n;
            const int rW = p.W >> p.res_up, rHW = rW * (p.H >> p.res_up);
            float bs[2][4];
            float2 r0[2][4], r1[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = crt * 32 + m * 16 + 4 * kk + i, rc = row < p.Cout ? row : p.Cout - 1;
                    bs[m][i] = bbase ? bbase[rc] : 0.f;
                    r0[m][i] = r1[m][i] = make_float2(0.f, 0.f);
                    if (rbase) {
                        const float* rp = rbase + ((long long)b * p.Cout + rc) * rHW;
                        if (p.res_up) {
                            const float r = rp[(y >> 1) * rW + (x >> 1)];
                            r0[m][i] = r1[m][i] = make_float2(r, r);
                        } else {
                            r0[m][i] = *reinterpret_cast<const float2*>(rp + y * rW + x);
                            r1[m][i] = *reinterpret_cast<const float2*>(rp + (y + 1) * rW + x);
                        }
                    }
                }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = crt * 32 + m * 16 + 4 * kk + i;
                    float M[16];
#pragma unroll
                    for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][m][i];
                    float y00, y01, y10, y11;
                    wino_out_transform(M, y00, y01, y10, y11);
                    y00 += bs[m][i] + r0[m][i].x; y01 += bs[m][i] + r0[m][i].y;
                    y10 += bs[m][i] + r1[m][i].x; y11 += bs[m][i] + r1[m][i].y;
                    if (eact != ACT_NONE) {
                        y00 = apply_act(y00, eact); y01 = apply_act(y01, eact);
                        y10 = apply_act(y10, eact); y11 = apply_act(y11, eact);
                    }
                    if (row < p.Cout && bval) {
                        float* op = obase + ((long long)b * p.Cout + row) * HW + y * p.W + x;
                        *reinterpret_cast<float2*>(op) = make_float2(y00, y01);
                        *reinterpret_cast<float2*>(op + p.W) = make_float2(y10, y11);
                    }
                }
#pragma unroll
            for (int x2 = 0; x2 < 16; ++x2)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[x2][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            after_epi = true;
        }
        if (!more) break;
        if (dyn) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) mbox[k & 1] = wino_claim_task(jnext, wino_xcc_id(), wino_claim_cs(p.ntasks), p.ntasks);
            __syncthreads();
            ct = tn;
            tn = __builtin_amdgcn_readfirstlane(mbox[k & 1]);
        } else {
            ct += G;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the block
}

// ================================================================================================================================
// SPADE gamma/beta conv (normalization.py:249-257) + the style convs conv_gamma / conv_beta (:117-153,172-173) + the fused ACE
// epilogue (:111-112,177-187; architecture.py:95) on the same Winograd machinery, over the BOUNDARY QUADS only (ace_sparse.h:
// a 2x2 quad is a boundary quad when one of its pixels has a non-uniform 5x5 label neighbourhood; the interior pass has
// written the others).
//   * GEMM rows: a row tile = 16 channels: rows 0-15 = their gamma rows, 16-31 = their beta rows, so a lane's accumulators
//     hold gamma and beta of the same (channel, pixel).
//   * K = the 128 SPADE hidden channels + (styled ACEs) 20 ONE-HOT label planes written behind them by the label-table kernel:
//     the style term  sum_t P[(sample, label(p + t)), t]  is the 3x3 conv of the one-hot map with per-SAMPLE weights P (the
//     style LUT), so it runs on the matrix cores as 5 more k-steps whose A images come from a per-sample buffer (wino_style_pack,
//     Winograd-transformed LUT) -- the 18 gathers per (pixel, 4-channel run) of the direct kernel's epilogue are gone.
//   * A block task = one spatial tile of 32 x 32 pixels x a PAIR of row tiles x up to 64 of its boundary quads: waves 0-3 / 4-7
//     take the two row tiles, wave (w & 3) the w-th group of 16 listed quads (B-fragment addresses are per lane anyway);
//     waves without quads only take part in the staging.  A tile with more than 64 boundary quads gets up to four tasks.
//     (Tiles of 32 x 16 left half of the waves idle on sparse levels -- about 30 boundary quads per tile at 512^2 on the
//     benchmark labels -- and a k-step costs nearly the same with four busy waves as with eight: 5.4 -> ms per C = 128 ACE.)
// Layout of the SPADE hidden activations both ACE kernels read (written by spade_hidden_wq, sean_kernels.hip): planes of H rows x
// wino_apitch(W) floats, image column x at x + WINO_AXOFF, zeros in the columns just left and right of the image.  The gather
// kernel fetches a quad's four patch rows (columns x - 1 .. x + 2) as ONE 16-byte LDS-DMA unit each; the offset of 32 pixels (one
// 128-byte line; rows stay 128-byte aligned) keeps the writer's runs of 64 pixels on whole lines (measured with an offset of 16
// pixels: the 512^2 label-table launches 480 -> 659 us, every run split over three lines).
constexpr int WINO_AXOFF = 32;
__host__ __device__ inline int wino_apitch(int W) { return W + 64; }
struct WinoAceParams {
    const float* actv;      // [B][K][H][wino_apitch(W)] (see above): K = 128 (+ 20 one-hot planes when wsty is set)
    const float* wpk;       // pack_wino_A image of the (gamma | beta) rows over the 128 hidden channels
    const float* wsty;      // [B][nrt][5][2048] per-sample style images, or null (unstyled ACE)
    float* out;             // [B][C][H][W]
    const float* x;         // [B][C][H >> x_up][W >> x_up]
    int x_up, act;
    int B, C, H, W;
    const float *bias_g, *bias_b, *bn_a, *bn_d, *nv;
    const float* noise;     // plane base of this ACE, sample stride noise_bstride, layout [W][H]
    long long noise_bstride;
    const uint8_t* qlist;   // [ntiles][8 TH] boundary quads of each tile of 32 x TH pixels (qy * 16 + qx), raster order
    int TH;                 // tile height 16 / 32 the lists were built with
    const int* qcnt;        // [ntiles]
    const unsigned* work;   // tile | row pair << 20 | part << 30 (part = which 64 of the tile's listed quads)
    const int* total;       // [0] = entries of `work`
    const float* zero;
    // gather mode (wino_ace_gather_kernel): the boundary quads of each sample as ONE list, tasks of 64 consecutive entries
    const unsigned* gq;     // [B][gq_cap]: y << 16 | x of the quad's first pixel
    const int* gq_n;        // [B] quads per sample
    int gq_cap;             // work entries in this mode: sample | chunk of 64 quads << 5 | row pair << 16
    unsigned* claim;        // gather mode: eight zeroed counters (dynamic task claiming, above) or null (static split)
    // gather mode, PATCH source (round 6): when a level is left with few boundary quads (straight-edge reduction, ace_sparse.h) their 4 x 4
    // patches are scattered 16-byte pieces of the hidden planes -- every piece pulls a whole 128-byte line through L1: the gather becomes
    // L1-fill bound at a quarter of its matrix rate.  spade_hidden_patch (sean_kernels.hip) then writes the patches of every chunk of 64
    // quads directly, in the stage layout [k-step][channel 4][patch row 4][slot 64][4 floats] (16 KB per k-step, contiguous), and the
    // patch DMA of a task becomes a flat copy.  chunk_base[b] = first chunk of sample b (chunk_base[B] = all), *patch_mode = 1 when the
    // level's chunks fit the buffer (decided on the device: wino_chunk_base); the same arithmetic on the same values: bit-identical.
    const float* patch;     // patch buffer, or null
    const int* chunk_base;  // [B + 1]
    const int* patch_mode;  // device flag
    int nrt, ntx, nty, K;   // set by the launcher
};
// Tile height 32 (16 x 16 quads) or 16 (16 x 8 quads), chosen per resolution level by the caller: a sparse tile must hold enough
// boundary quads to fill the block's waves (512^2 on the benchmark labels: about 30 boundary quads per 32 x 16 tile, half of the
// waves idle, and a k-step costs nearly the same with four busy waves as with eight: 13.8 -> 9.6 ms for the level's three ACEs
// with the taller tile), but the taller patch costs 20 instead of 12 KB of DMA per k-step and one ring stage (256^2: 8.1 vs 9.6).
template <int TH_>
struct WaCfg {
    static constexpr int TH = TH_, PROWS = TH + 2, NQ = 8 * TH;              // quads per tile
    static constexpr int PS = TH == 32 ? 1248 : wino::PS;                     // plane stride (floats), = 32 mod 64
    static constexpr int NPD = TH == 32 ? 10 : 6, PDW = NPD * 512;            // patch DMA instructions per thread per k-step
    static constexpr int NST = TH == 32 ? 4 : 5;                              // ring depth
    static constexpr int ADW = 2 * wino::ADW, SDW = PDW + ADW, NLD = NPD + 2;
    static constexpr int LDS_BYTES = NST * SDW * 4;                           // 144 KB / 140 KB
};

template <int TH_>
__global__ __launch_bounds__(512, 1) void wino_ace_kernel(const WinoAceParams p) {
    using namespace wino;
    using Cfg = WaCfg<TH_>;
    constexpr int WA_TH = Cfg::TH, WA_PROWS = Cfg::PROWS, WA_PS = Cfg::PS, WA_NPD = Cfg::NPD, WA_PDW = Cfg::PDW, WA_NST = Cfg::NST;
    constexpr int WA_SDW = Cfg::SDW, WA_NLD = Cfg::NLD, WA_NQ = Cfg::NQ;
    constexpr int AHEAD = WA_NST - 1;                          // k-steps the issue side runs ahead of the consumers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int G = gridDim.x;
    const int ntasks = p.total[0];
    const int lb = xcd_remap(blockIdx.x, G);      // the tasks of a tile (row pairs x parts: consecutive entries) run on ONE XCD at the same time
    if (lb >= ntasks) return;
    const int mytasks = (ntasks - lb + G - 1) / G;
    const int nks = 32, nk = nks + (p.wsty ? 5 : 0);
    const int HW = p.H * p.W;
    const int AP = wino_apitch(p.W), APL = p.H * AP;           // pitch / plane size of the hidden activations (floats)
    constexpr unsigned SB = WA_SDW * 4, RING = WA_NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    // ---- issue side ------------------------------------------------------------------------------------------------------------
    // The k-step is BRANCH-FREE (measured: every conditional inside it -- around an MFMA group, an operand read, the choice of the A
    // source -- cost far more than its instruction count: tools/wino_ace_bench.hip): the transitions of the issue side (hidden ->
    // style A images -> next task) happen BETWEEN straight runs of k-steps, see the task loop.
    unsigned voff[WA_NPD];
    const unsigned va = (unsigned)tid * 16u;
    wino_u32x4 d_in, d_h0, d_h1, d_s0, d_s1, dA0, dA1;
    unsigned so_in = 0, so_a = 0;
    auto issue_task = [&](int t) {
        const unsigned wk = p.work[t];
        const int tile = wk & 0xFFFFF, pair = (wk >> 20) & 0x3FF;
        const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, ib = tile / (p.ntx * p.nty);
        const int y0 = ty * WA_TH - 1, x0 = tx * TW - 1;
#pragma unroll
        for (int i = 0; i < WA_NPD; ++i) {
            const int e = i * 512 + tid;
            const int k4 = e / WA_PS, rem = e - k4 * WA_PS;
            const int py = rem / PWP, px = rem - py * PWP;
            const int y = y0 + py, x = x0 + px;
            const bool ok = k4 < 4 && rem < WA_PROWS * PWP && px < TW + 2 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            voff[i] = ok ? (unsigned)(k4 * APL + y * AP + x + WINO_AXOFF) * 4u : 0x80000000u;
        }
        const int r0 = 2 * pair, r1 = 2 * pair + 1 < p.nrt ? 2 * pair + 1 : 2 * pair;      // (odd row-tile count: the last pair repeats)
        d_in = wino_rsrc(p.actv + (long long)ib * p.K * APL, (unsigned)p.K * APL * 4u);
        d_h0 = wino_rsrc(p.wpk + (long long)r0 * nks * 2048, (unsigned)nks * 8192u);
        d_h1 = wino_rsrc(p.wpk + (long long)r1 * nks * 2048, (unsigned)nks * 8192u);
        if (p.wsty) {
            d_s0 = wino_rsrc(p.wsty + ((long long)ib * p.nrt + r0) * 5 * 2048, 5u * 8192u);
            d_s1 = wino_rsrc(p.wsty + ((long long)ib * p.nrt + r1) * 5 * 2048, 5u * 8192u);
        }
        dA0 = d_h0;
        dA1 = d_h1;
        so_in = 0;
        so_a = 0;
    };
    unsigned islot = lds0;                                     // ring slot the next issue fills
    // the LDS-DMA instructions of a k-step, spread over the eight MFMA groups so that they issue in the shadow of the matrix pipe
    auto issue_group = [&](auto gt) {
        constexpr int g = decltype(gt)::value;
        if constexpr (g < 6) {
            constexpr int lo = g * WA_NPD / 6, hi = (g + 1) * WA_NPD / 6;
            const unsigned wb = islot + (unsigned)wave * 256u;
#pragma unroll
            for (int i = lo; i < hi; ++i) wino_dma4(voff[i], d_in, so_in, wb + (unsigned)i * 2048u);
        } else if constexpr (g == 6) {
            wino_dma16(va, dA0, so_a, islot + WA_PDW * 4u + (unsigned)wave * 1024u);
        } else {
            wino_dma16(va, dA1, so_a, islot + WA_PDW * 4u + (unsigned)wave * 1024u + ADW * 4u);
            islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
            so_in += 16u * (unsigned)APL;
            so_a += 8192u;
        }
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[x][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this wave's share of a task: quad group (wave & 3) + 4 * part of the tile's list, row tile 2 * pair + (wave >> 2)
    struct Ctx { int tile, rt, qy, qx, boff; bool active, valid; };
    auto task_ctx = [&](int t) {
        Ctx c;
        const unsigned wk = p.work[t];
        c.tile = wk & 0xFFFFF;
        const int pair = (wk >> 20) & 0x3FF, part = wk >> 30;
        const int cnt = p.qcnt[c.tile];
        const int grp = (wave & 3) + 4 * part;
        c.rt = 2 * pair + (wave >> 2);
        c.active = grp * 16 < cnt && c.rt < p.nrt;
        const int qi = grp * 16 + n;
        c.valid = c.active && qi < cnt;
        const int q = p.qlist[(long long)c.tile * WA_NQ + (qi < cnt ? qi : (cnt > 0 ? cnt - 1 : 0))];
        c.qy = q >> 4;
        c.qx = q & 15;
        c.boff = kk * WA_PS + (2 * c.qy) * PWP + 2 * c.qx;
        return c;
    };
    Ctx cur = task_ctx(lb);
    const int aoff = WA_PDW + (wave >> 2) * ADW;               // this wave's A image inside a stage (floats)
    auto load_raw = [&](unsigned slot, int boff, float (&d)[4][4]) {
        const float* sp = reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + boff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 lo = *reinterpret_cast<const float2*>(sp + i * PWP), hi = *reinterpret_cast<const float2*>(sp + i * PWP + 2);
            d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
        }
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + aoff) + lane; };

    issue_task(lb);
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_group(WInt<0>{}); issue_group(WInt<1>{}); issue_group(WInt<2>{}); issue_group(WInt<3>{});
        issue_group(WInt<4>{}); issue_group(WInt<5>{}); issue_group(WInt<6>{}); issue_group(WInt<7>{});
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WA_NLD * (WA_NST - 3)) : "memory");
    __syncthreads();
    // B fragments of the current k-step (transformed patch of this lane's (quad, channel)); overwritten IN PLACE by the next k-step's
    // as the MFMA groups release them.  F: A fragments, group g reads F[g & 3], the read for group g + 2 is issued at group g.
    float v[16], t3[4];
    f32x4 F[4];
    {
        float d[4][4];
        load_raw(lds0, cur.boff, d);
        wino_in_transform(d, v);                               // (v[12..15] are written again, from t3, by the first k-step)
#pragma unroll
        for (int j = 0; j < 4; ++j) t3[j] = d[1][j] - d[3][j];
        const f32x4* ap = a_ptr(lds0);
        F[0] = ap[0];
        F[1] = ap[64];
        F[2] = F[3] = F[0];
    }
    unsigned rslot = lds0;
    // One k-step (4 input channels; eight groups of four MFMAs) + the fetch and input transform of the next k-step's operands
    // (boff_pre = the lane's patch origin in THAT k-step's task).  Every wave runs it, with or without quads in the task (a wave
    // without quads sits on a SIMD of its own with its row-tile twin -- waves w and w + 4 share the quad group -- so its MFMAs on
    // clamped operands delay nobody, and one code path keeps the accumulators out of scratch); the epilogue is what is predicated.
    auto kstep = [&](int boff_pre) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WA_NLD * (WA_NST - 3)) : "memory");
        __syncthreads();
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const f32x4* apn = a_ptr(nslot);                       // (k-step q + 1 was verified by the barrier above)
        float dn[4][4], t[4][4];
        // eighth e of the k-step: the A-fragment read for eighth e + 2 (F is a ring of four), four MFMAs, a piece of the next k-step's
        // input transform behind the MFMAs that read the old values, a piece of the k-step's LDS-DMA issue
        auto g_loads = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            if constexpr (g < 6) F[(g + 2) & 3] = ap[(g + 2) * 64];
            else F[(g + 2) & 3] = apn[(g - 6) * 64];
            if constexpr (g == 0) load_raw(nslot, boff_pre, dn);
        };
        auto g_math = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            const f32x4 c = F[g & 3];
            acc[2 * g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, v[2 * g], acc[2 * g][0], 0, 0, 0);
            acc[2 * g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, v[2 * g], acc[2 * g][1], 0, 0, 0);
            acc[2 * g + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, v[2 * g + 1], acc[2 * g + 1][0], 0, 0, 0);
            acc[2 * g + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, v[2 * g + 1], acc[2 * g + 1][1], 0, 0, 0);
            if constexpr (g == 0) {                            // row 3 of the transform the previous k-step computed: its old values were
                v[12] = t3[0] - t3[2];                         //   read by that k-step's last two eighths (a write behind them would have
                v[13] = t3[1] + t3[2];                         //   waited for their MFMAs to start)
                v[14] = t3[2] - t3[1];
                v[15] = t3[1] - t3[3];
            }
            if constexpr (g == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = dn[0][j] - dn[2][j];
                    t[1][j] = dn[1][j] + dn[2][j];
                    t[2][j] = dn[2][j] - dn[1][j];
                    t[3][j] = dn[1][j] - dn[3][j];
                }
            }
            if constexpr (g >= 4 && g < 7) {
                constexpr int i = g - 4;                       // v[4 i ..]: read by the eighths 2 i, 2 i + 1 < g of this k-step
                v[4 * i + 0] = t[i][0] - t[i][2];
                v[4 * i + 1] = t[i][1] + t[i][2];
                v[4 * i + 2] = t[i][2] - t[i][1];
                v[4 * i + 3] = t[i][1] - t[i][3];
            }
            if constexpr (g == 7) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t3[j] = t[3][j];
            }
        };
#if WINO_NG == 8
        auto group = [&](auto gt) {
            g_loads(gt);
            __builtin_amdgcn_sched_barrier(0);                 // the LDS reads of the group after next first: two groups of MFMA time are their latency
            g_math(gt);
            __builtin_amdgcn_sched_barrier(0);
            issue_group(gt);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{});
        group(WInt<4>{}); group(WInt<5>{}); group(WInt<6>{}); group(WInt<7>{});
#else
        auto group = [&](auto g0, auto g1) {                   // four groups of eight MFMAs
            g_loads(g0); g_loads(g1);
            __builtin_amdgcn_sched_barrier(0);                 // the LDS reads of the next group first: this group's MFMA time is their latency
            g_math(g0); g_math(g1);
            __builtin_amdgcn_sched_barrier(0);
            issue_group(g0); issue_group(g1);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}, WInt<1>{}); group(WInt<2>{}, WInt<3>{}); group(WInt<4>{}, WInt<5>{}); group(WInt<6>{}, WInt<7>{});
#endif
        rslot = nslot;
    };
    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        const bool more = k + 1 < mytasks;
        Ctx nxt = cur;
        if (more) nxt = task_ctx(ct + G);                       // (its list entry is in flight during this task's k-steps)
        const int tnext = more ? ct + G : ct;                   // (past the end: this task again -- never read; keeps the vmcnt counting uniform)
        // consumer k-step c issues k-step c + AHEAD: hidden channels while c + AHEAD < 32, then the style images, then the next task
        for (int c = 0; c < nks - AHEAD; ++c) kstep(cur.boff);
        if (nk > nks) {
            dA0 = d_s0;
            dA1 = d_s1;
            so_a = 0;
            for (int c = nks - AHEAD; c < nk - AHEAD; ++c) kstep(cur.boff);
        }
        issue_task(tnext);
        for (int c = nk - AHEAD; c < nk; ++c) kstep(c == nk - 1 ? nxt.boff : cur.boff);      // the last k-step's prefetch belongs to the next task
        // ---- ACE epilogue of this wave's (quads, row tile): every load first (channels clamped, so no load sits behind a branch: the
        //      per-channel load -> use chains of the first version cost four memory round trips per task), then the output transforms
        //      and the modulation, then the stores ------------------------------------------------------------------------------
        if (cur.active) {
            const int tile = cur.tile;
            const int tx = tile % p.ntx, ty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
            const int y = ty * WA_TH + 2 * cur.qy, x = tx * TW + 2 * cur.qx;
            const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
            const float* nzp = p.noise + (long long)b * p.noise_bstride + (long long)x * p.H + y;      // plane layout [W][H]
            const float2 nz0 = *reinterpret_cast<const float2*>(nzp), nz1 = *reinterpret_cast<const float2*>(nzp + p.H);
            // nz0 = (y, x), (y + 1, x);  nz1 = (y, x + 1), (y + 1, x + 1)
            const int c0 = cur.rt * 16 + 4 * kk, c0c = c0 < p.C ? c0 : p.C - 4;                          // (C % 4 == 0: conv_wino_ace)
            const float4 pg = *reinterpret_cast<const float4*>(p.bias_g + c0c), pb = *reinterpret_cast<const float4*>(p.bias_b + c0c);
            const float4 pa = *reinterpret_cast<const float4*>(p.bn_a + c0c), pd = *reinterpret_cast<const float4*>(p.bn_d + c0c);
            const float4 pn = *reinterpret_cast<const float4*>(p.nv + c0c);
            float2 xr0[4], xr1[4];
            const float* xp0 = p.x + ((long long)b * p.C + c0c) * xHW;
            if (p.x_up) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = xp0[(long long)i * xHW + (y >> 1) * xW + (x >> 1)];
                    xr0[i] = xr1[i] = make_float2(xv, xv);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xr0[i] = *reinterpret_cast<const float2*>(xp0 + (long long)i * xHW + y * xW + x);
                    xr1[i] = *reinterpret_cast<const float2*>(xp0 + (long long)i * xHW + (y + 1) * xW + x);
                }
            }
            const float g_[4] = {pg.x, pg.y, pg.z, pg.w}, b_[4] = {pb.x, pb.y, pb.z, pb.w};
            const float a_[4] = {pa.x, pa.y, pa.z, pa.w}, d_[4] = {pd.x, pd.y, pd.z, pd.w}, n_[4] = {pn.x, pn.y, pn.z, pn.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i;
                float M[16], g00, g01, g10, g11, e00, e01, e10, e11;
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][0][i];
                wino_out_transform(M, g00, g01, g10, g11);
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][1][i];
                wino_out_transform(M, e00, e01, e10, e11);
                const float gb = 1.f + g_[i], bb = b_[i];
                float o00 = (a_[i] * xr0[i].x + n_[i] * nz0.x + d_[i]) * (gb + g00) + (bb + e00);
                float o01 = (a_[i] * xr0[i].y + n_[i] * nz1.x + d_[i]) * (gb + g01) + (bb + e01);
                float o10 = (a_[i] * xr1[i].x + n_[i] * nz0.y + d_[i]) * (gb + g10) + (bb + e10);
                float o11 = (a_[i] * xr1[i].y + n_[i] * nz1.y + d_[i]) * (gb + g11) + (bb + e11);
                if (p.act != ACT_NONE) {
                    o00 = apply_act(o00, p.act); o01 = apply_act(o01, p.act);
                    o10 = apply_act(o10, p.act); o11 = apply_act(o11, p.act);
                }
                if (cur.valid && c < p.C) {
                    float* op = p.out + ((long long)b * p.C + c) * HW + y * p.W + x;
                    *reinterpret_cast<float2*>(op) = make_float2(o00, o01);
                    *reinterpret_cast<float2*>(op + p.W) = make_float2(o10, o11);
                }
            }
        }
#pragma unroll
        for (int x2 = 0; x2 < 16; ++x2)                         // (waves without quads in the task accumulated products of clamped operands)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[x2][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's loads / stores share the counter with the ring: drain once per task
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the block
}

// ---- gather mode ------------------------------------------------------------------------------------------------------------------
// The tile kernel above stages the patch of a whole tile of 32 x TH pixels for the (at most 64) boundary quads it holds: on sparse
// levels most of that patch is never read, a tile rarely holds a multiple of 16 quads (256^2 on the benchmark labels: 56 -> 86 % of
// the MFMA slots), and the taller tiles that the sparse levels need cost ring depth (TH = 32: four stages of 36 KB).  Here a task is 64
// CONSECUTIVE entries of the sample's list of boundary quads x a pair of row tiles, wherever those quads lie: the LDS-DMA fetches
// each quad's own 4 x 4 patch (per-lane source offsets were already the access pattern, so a gather costs the same eight
// instructions per thread and k-step), stage layout [channel][patch row][quad slot][4 floats], 16 KB per k-step whatever the
// level; a lane's four B rows are four conflict-free ds_read_b128.  Every task but the last of a sample fills all 64 slots.
// Same arithmetic per quad as the tile kernel: bit-identical results (tests/test_hip_wino.py).
namespace winog {
constexpr int NPD = 2, PDW = 4096, NST = 5, ADW2 = 2 * wino::ADW, SDW = PDW + ADW2, NLD = NPD + 2;
constexpr int LDS_BYTES = NST * SDW * 4;                         // 128 KB
}  // namespace winog

template <int DUMMY>
__global__ __launch_bounds__(512, 1) void wino_ace_gather_kernel(const WinoAceParams p) {
    using namespace wino;
    constexpr int WA_NST = winog::NST, WA_NLD = winog::NLD, WA_PDW = winog::PDW;
    constexpr int AHEAD = WA_NST - 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int G = gridDim.x;
    const int ntasks = p.total[0];
    // static split: block lb takes tasks lb, lb + G, ... (the row pairs of a chunk of quads are consecutive entries: one XCD, the same
    // time); dynamic: tasks claimed from the XCD's counter (wino_claim_task).  This kernel's ring fills all 160 KB of LDS, so the
    // claimed task travels from wave 0 to the other waves through device memory: p.claim[16 + 2 blockIdx + parity], written after
    // the epilogue's drain and read eight k-steps into the next task by a scalar load that bypasses the scalar cache (wave 0's
    // counted waits in between retire the store; both ends sit on one CU, L2 is their coherence point).  The first two tasks are
    // claimed while the ring is still empty and pass through its first words.
    const bool dyn = p.claim != nullptr && ntasks >= 8 * G;       // (a block holds two claimed tasks: only where a block gets many)
    int lb, tn;                                    // first task, the task after it (ntasks: none)
    if (dyn) {
        int* mb = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const int xcc = wino_xcc_id(), ccs = wino_claim_cs(ntasks);
            const unsigned j0 = __hip_atomic_fetch_add(p.claim + xcc, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mb[0] = wino_claim_task(j0, xcc, ccs, ntasks);
            mb[1] = wino_claim_task(j0 + 1u, xcc, ccs, ntasks);
        }
        __syncthreads();
        lb = __builtin_amdgcn_readfirstlane(mb[0]);
        tn = __builtin_amdgcn_readfirstlane(mb[1]);
        __syncthreads();
    } else {
        lb = xcd_remap(blockIdx.x, G);
        tn = lb + G < ntasks ? lb + G : ntasks;
    }
    if (lb >= ntasks) return;
    const int nks = 32, nk = nks + (p.wsty ? 5 : 0);
    const int HW = p.H * p.W;
    constexpr unsigned SB = winog::SDW * 4, RING = WA_NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    // ---- issue side: lane = quad slot; wave w fetches patch row (w & 3) of channels (w >> 2) and 2 + (w >> 2) of the k-step: a row of
    //      a quad's patch -- columns x - 1 .. x + 2 -- is ONE 16-byte unit of the padded plane (4-byte aligned), a wave instruction
    //      fills one [64 slots][4 floats] line of the stage.  Two patch + two A instructions per wave and k-step (the first version
    //      issued eight 4-byte patch DMAs per thread: ten vector-memory instructions per k-step next to 32 MFMAs). -----------------
    const int AP = wino_apitch(p.W), APL = p.H * AP;
    const int irow = wave & 3;
    unsigned vb = 0x80000000u;                                  // byte offset of this lane's unit (row irow of its slot's patch) in a plane
    const unsigned va = (unsigned)tid * 16u;
    wino_u32x4 d_in, d_h0, d_h1, d_s0, d_s1, dA0, dA1;
    unsigned so_in = 0, so_a = 0;
    const bool from_patch = p.patch != nullptr && p.patch_mode != nullptr && __builtin_amdgcn_readfirstlane(*p.patch_mode) == 1;
    const unsigned APL4 = from_patch ? 4096u : (unsigned)APL * 4u;      // bytes from one channel's patch rows to the next one's
    auto task_quad = [&](int t, int slot) {                       // entry `slot` of task t's chunk (clamped to the sample's list)
        const unsigned wk = p.work[t];
        const int b = wk & 31, chunk = (wk >> 5) & 2047;
        const int nq = p.gq_n[b];
        int qi = chunk * 64 + slot;
        qi = qi < nq ? qi : nq - 1;
        return p.gq[(long long)b * p.gq_cap + qi];
    };
    auto issue_task = [&](int t, unsigned q) {                    // q = task_quad(t, lane), fetched a task ahead
        const unsigned wk = p.work[t];
        const int ib = wk & 31, pair = wk >> 16;
        const int y = (int)(q >> 16) - 1 + irow, x = (int)(q & 0xFFFFu);      // (x - 1 + WINO_AXOFF: always inside the padded row)
        const int r0 = 2 * pair, r1 = 2 * pair + 1 < p.nrt ? 2 * pair + 1 : 2 * pair;      // (odd row-tile count: the last pair repeats)
        if (from_patch) {        // the chunk's pre-gathered patches: channel c of k-step s at (4 s + c) * 4 KB, row irow, this lane's slot
            const int chunk = (wk >> 5) & 2047;
            vb = (unsigned)irow * 1024u + (unsigned)lane * 16u;
            d_in = wino_rsrc(p.patch + (long long)(p.chunk_base[ib] + chunk) * p.K * 1024, (unsigned)p.K * 4096u);
        } else {
            vb = (unsigned)y < (unsigned)p.H ? (unsigned)(y * AP + x + WINO_AXOFF - 1) * 4u : 0x80000000u;
            d_in = wino_rsrc(p.actv + (long long)ib * p.K * APL, (unsigned)p.K * APL * 4u);
        }
        d_h0 = wino_rsrc(p.wpk + (long long)r0 * nks * 2048, (unsigned)nks * 8192u);
        d_h1 = wino_rsrc(p.wpk + (long long)r1 * nks * 2048, (unsigned)nks * 8192u);
        if (p.wsty) {
            d_s0 = wino_rsrc(p.wsty + ((long long)ib * p.nrt + r0) * 5 * 2048, 5u * 8192u);
            d_s1 = wino_rsrc(p.wsty + ((long long)ib * p.nrt + r1) * 5 * 2048, 5u * 8192u);
        }
        dA0 = d_h0;
        dA1 = d_h1;
        so_in = 0;
        so_a = 0;
    };
    unsigned islot = lds0;
    // stage layout [channel][patch row][slot][4 floats]: line (c, row) at byte (4 c + row) * 1024
    auto issue_group = [&](auto gt) {
        constexpr int g = decltype(gt)::value;
        if constexpr (g == 0) wino_dma16(vb, d_in, so_in + (unsigned)(wave >> 2) * APL4, islot + (unsigned)(4 * (wave >> 2) + irow) * 1024u);
        if constexpr (g == 2) wino_dma16(vb, d_in, so_in + (unsigned)(2 + (wave >> 2)) * APL4, islot + (unsigned)(4 * (2 + (wave >> 2)) + irow) * 1024u);
        if constexpr (g == 4) wino_dma16(va, dA0, so_a, islot + WA_PDW * 4u + (unsigned)wave * 1024u);
        if constexpr (g == 6) {
            wino_dma16(va, dA1, so_a, islot + WA_PDW * 4u + (unsigned)wave * 1024u + ADW * 4u);
            islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
            so_in += 4u * APL4;
            so_a += 8192u;
        }
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[16][2];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[x][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int cslot = (wave & 3) * 16 + n;                        // this lane's quad slot; row tile 2 * pair + (wave >> 2)
    struct Ctx { int b, rt, y, x; bool active, valid; };
    auto task_ctx = [&](int t) {
        Ctx c;
        const unsigned wk = p.work[t];
        c.b = wk & 31;
        const int chunk = (wk >> 5) & 2047, pair = wk >> 16;
        const int nq = p.gq_n[c.b];
        c.rt = 2 * pair + (wave >> 2);
        c.active = chunk * 64 + (wave & 3) * 16 < nq && c.rt < p.nrt;
        c.valid = c.active && chunk * 64 + cslot < nq;
        const unsigned q = task_quad(t, cslot);
        c.y = (int)(q >> 16);
        c.x = (int)(q & 0xFFFFu);
        return c;
    };
    Ctx cur = task_ctx(lb);
    const int aoff = WA_PDW + (wave >> 2) * ADW;
    const int boff = kk * 1024 + cslot * 4;                       // [channel][patch row][slot][4]
    auto load_raw = [&](unsigned slot, float (&d)[4][4]) {
        const float* sp = reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + boff;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(sp + i * 256);
            d[i][0] = r.x; d[i][1] = r.y; d[i][2] = r.z; d[i][3] = r.w;
        }
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem) + (slot - lds0) / 4 + aoff) + lane; };

    issue_task(lb, task_quad(lb, lane));
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) {
        issue_group(WInt<0>{}); issue_group(WInt<1>{}); issue_group(WInt<2>{}); issue_group(WInt<3>{});
        issue_group(WInt<4>{}); issue_group(WInt<5>{}); issue_group(WInt<6>{}); issue_group(WInt<7>{});
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WA_NLD * (WA_NST - 3)) : "memory");
    __syncthreads();
    float v[16], t3[4];
    f32x4 F[4];
    {
        float d[4][4];
        load_raw(lds0, d);
        wino_in_transform(d, v);                               // (v[12..15] are written again, from t3, by the first k-step)
#pragma unroll
        for (int j = 0; j < 4; ++j) t3[j] = d[1][j] - d[3][j];
        const f32x4* ap = a_ptr(lds0);
        F[0] = ap[0];
        F[1] = ap[64];
        F[2] = F[3] = F[0];
    }
    unsigned rslot = lds0;
    auto kstep = [&]() {                                       // (see wino_ace_kernel: branch-free, every wave runs it)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WA_NLD * (WA_NST - 3)) : "memory");
        __syncthreads();
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const f32x4* apn = a_ptr(nslot);
        float dn[4][4], t[4][4];
        auto g_loads = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            if constexpr (g < 6) F[(g + 2) & 3] = ap[(g + 2) * 64];
            else F[(g + 2) & 3] = apn[(g - 6) * 64];
            if constexpr (g == 0) load_raw(nslot, dn);
        };
        auto g_math = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            const f32x4 c = F[g & 3];
            acc[2 * g][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, v[2 * g], acc[2 * g][0], 0, 0, 0);
            acc[2 * g][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, v[2 * g], acc[2 * g][1], 0, 0, 0);
            acc[2 * g + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, v[2 * g + 1], acc[2 * g + 1][0], 0, 0, 0);
            acc[2 * g + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, v[2 * g + 1], acc[2 * g + 1][1], 0, 0, 0);
            if constexpr (g == 0) {
                v[12] = t3[0] - t3[2];
                v[13] = t3[1] + t3[2];
                v[14] = t3[2] - t3[1];
                v[15] = t3[1] - t3[3];
            }
            if constexpr (g == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = dn[0][j] - dn[2][j];
                    t[1][j] = dn[1][j] + dn[2][j];
                    t[2][j] = dn[2][j] - dn[1][j];
                    t[3][j] = dn[1][j] - dn[3][j];
                }
            }
            if constexpr (g >= 4 && g < 7) {
                constexpr int i = g - 4;
                v[4 * i + 0] = t[i][0] - t[i][2];
                v[4 * i + 1] = t[i][1] + t[i][2];
                v[4 * i + 2] = t[i][2] - t[i][1];
                v[4 * i + 3] = t[i][1] - t[i][3];
            }
            if constexpr (g == 7) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t3[j] = t[3][j];
            }
        };
        auto group = [&](auto gt) {
            g_loads(gt);
            __builtin_amdgcn_sched_barrier(0);
            g_math(gt);
            __builtin_amdgcn_sched_barrier(0);
            issue_group(gt);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{});
        group(WInt<4>{}); group(WInt<5>{}); group(WInt<6>{}); group(WInt<7>{});
        rslot = nslot;
    };
    constexpr int KREAD = 8;                                   // k-steps into a task at which the next task's id is read
    for (int k = 0, ct = lb;; ++k) {
        for (int c = 0; c < KREAD; ++c) kstep();
        if (k > 0) {
            if (dyn) {
                unsigned v;
                do v = (unsigned)wino_mail_read(p.claim + 16 + 2 * blockIdx.x + ((k - 1) & 1));
                while ((v >> 21) != (unsigned)(k & 0x7FF));
                tn = (int)(v & 0x1FFFFFu);
            }
            else tn = ct + G < ntasks ? ct + G : ntasks;
        }
        const bool more = tn < ntasks;
        const int tnext = more ? tn : ct;                       // (past the end: this task again -- never read; keeps the vmcnt counting uniform)
        Ctx nxt = cur;
        if (more) nxt = task_ctx(tnext);                        // (in flight during this task's remaining k-steps)
        const unsigned qnext = task_quad(tnext, lane);
        for (int c = KREAD; c < nks - AHEAD; ++c) kstep();
        if (nk > nks) {
            dA0 = d_s0;
            dA1 = d_s1;
            so_a = 0;
            for (int c = nks - AHEAD; c < nk - AHEAD; ++c) kstep();
        }
        issue_task(tnext, qnext);
        for (int c = nk - AHEAD; c < nk; ++c) kstep();
        unsigned jnext = 0;                                     // dynamic claiming: the task after next, answered during the epilogue
        if (dyn && more && tid == 0) jnext = __hip_atomic_fetch_add(p.claim + wino_xcc_id(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- ACE epilogue (as wino_ace_kernel) ---------------------------------------------------------------------------------
        if (cur.active) {
            const int b = cur.b, y = cur.y, x = cur.x;
            const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
            const float* nzp = p.noise + (long long)b * p.noise_bstride + (long long)x * p.H + y;      // plane layout [W][H]
            const float2 nz0 = *reinterpret_cast<const float2*>(nzp), nz1 = *reinterpret_cast<const float2*>(nzp + p.H);
            const int c0 = cur.rt * 16 + 4 * kk, c0c = c0 < p.C ? c0 : p.C - 4;
            const float4 pg = *reinterpret_cast<const float4*>(p.bias_g + c0c), pb = *reinterpret_cast<const float4*>(p.bias_b + c0c);
            const float4 pa = *reinterpret_cast<const float4*>(p.bn_a + c0c), pd = *reinterpret_cast<const float4*>(p.bn_d + c0c);
            const float4 pn = *reinterpret_cast<const float4*>(p.nv + c0c);
            float2 xr0[4], xr1[4];
            const float* xp0 = p.x + ((long long)b * p.C + c0c) * xHW;
            if (p.x_up) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = xp0[(long long)i * xHW + (y >> 1) * xW + (x >> 1)];
                    xr0[i] = xr1[i] = make_float2(xv, xv);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xr0[i] = *reinterpret_cast<const float2*>(xp0 + (long long)i * xHW + y * xW + x);
                    xr1[i] = *reinterpret_cast<const float2*>(xp0 + (long long)i * xHW + (y + 1) * xW + x);
                }
            }
            const float g_[4] = {pg.x, pg.y, pg.z, pg.w}, b_[4] = {pb.x, pb.y, pb.z, pb.w};
            const float a_[4] = {pa.x, pa.y, pa.z, pa.w}, d_[4] = {pd.x, pd.y, pd.z, pd.w}, n_[4] = {pn.x, pn.y, pn.z, pn.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i;
                float M[16], g00, g01, g10, g11, e00, e01, e10, e11;
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][0][i];
                wino_out_transform(M, g00, g01, g10, g11);
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) M[xi] = acc[xi][1][i];
                wino_out_transform(M, e00, e01, e10, e11);
                const float gb = 1.f + g_[i], bb = b_[i];
                float o00 = (a_[i] * xr0[i].x + n_[i] * nz0.x + d_[i]) * (gb + g00) + (bb + e00);
                float o01 = (a_[i] * xr0[i].y + n_[i] * nz1.x + d_[i]) * (gb + g01) + (bb + e01);
                float o10 = (a_[i] * xr1[i].x + n_[i] * nz0.y + d_[i]) * (gb + g10) + (bb + e10);
                float o11 = (a_[i] * xr1[i].y + n_[i] * nz1.y + d_[i]) * (gb + g11) + (bb + e11);
                if (p.act != ACT_NONE) {
                    o00 = apply_act(o00, p.act); o01 = apply_act(o01, p.act);
                    o10 = apply_act(o10, p.act); o11 = apply_act(o11, p.act);
                }
                if (cur.valid && c < p.C) {
                    float* op = p.out + ((long long)b * p.C + c) * HW + y * p.W + x;
                    *reinterpret_cast<float2*>(op) = make_float2(o00, o01);
                    *reinterpret_cast<float2*>(op + p.W) = make_float2(o10, o11);
                }
            }
        }
#pragma unroll
        for (int x2 = 0; x2 < 16; ++x2)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[x2][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!more) break;
        // (tagged with the iteration: the reader spins until it sees THIS iteration's word -- the counted waits make that the first read)
        if (dyn && tid == 0)
            p.claim[16 + 2 * blockIdx.x + (k & 1)] = (unsigned)wino_claim_task(jnext, wino_xcc_id(), wino_claim_cs(ntasks), ntasks) | (unsigned)((k + 1) & 0x7FF) << 21;
        cur = nxt;
        ct = tn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the block
}

inline bool wino_supported(int H, int W, int Cin) { return H % wino::TH == 0 && W % wino::TW == 0 && Cin % 8 == 0 && H >= 16; }
// 16 x 16 images (the generator's head block at 512^2): pairs of samples share a tile
// (an odd batch leaves the last tile's second half empty: its patch reads fall behind the tensor's end and return zeros, its stores are skipped)
inline bool wino_supported_pair16(int B, int H, int W, int Cin) { return H == 16 && W == 16 && B >= 1 && Cin % 8 == 0; }

inline void wino_fill_launch(WinoParams& p) {
    p.nrt = (p.Cout + 31) / 32;
    p.pair16 = (p.W == 16 && p.H == 16) ? 1 : 0;
    p.ntx = p.pair16 ? 1 : p.W / wino::TW;
    p.nty = p.H / wino::TH;
    p.ntiles = (p.pair16 ? (p.B + 1) / 2 : p.B) * p.ntx * p.nty;
    p.ntasks = p.ntiles * p.nrt;
    p.nks = p.Cin / 4;
    p.rb = p.nrt >= 4 ? 4 : p.nrt;
    p.tbk = 32 / p.rb;
}

hipError_t conv_wino_plain(WinoParams p, hipStream_t s);     // conv_inst_wino.hip (chooses the K split and runs the reduce pass itself)
hipError_t conv_wino_ace(WinoAceParams p, hipStream_t s);
// wsty[b][rt][s][idx][lane][4] <- Winograd transform of the style LUT lut[(b*19 + j)][tap][gamma|beta][C] (exact-f32 layout)
hipError_t wino_style_pack(const float* lut, float* wsty, int B, int C, hipStream_t s);
// boundary quads of every 32 x TH tile (from the interior map u5 of ace_classify) and the block tasks of conv_wino_ace
// (u5 == nullptr: every quad is a boundary quad -- levels without the interior reduction); pcnt: boundary pixels per tile; total: 8 ints
hipError_t wino_quad_lists(const uint8_t* u5, uint8_t* qlist, int* qcnt, int* pcnt, int B, int H, int W, int TH, hipStream_t s);
hipError_t wino_ace_worklist(const int* qcnt, const int* pcnt, int ntiles, int nrt, unsigned* work, int* total, hipStream_t s);
// gather mode: the per-tile lists (tiles of 32 x 16) -> one list per sample gq[b][gq_cap] (tile order), counts gq_n[b], scratch
// qoff[ntiles]; then the tasks (sample | chunk << 5 | row pair << 16; chunk-major) and the statistics of wino_ace_worklist
hipError_t wino_gather_lists(const uint8_t* qlist, const int* qcnt, int* qoff, unsigned* gq, int* gq_n, int gq_cap, int B, int H, int W, hipStream_t s);
hipError_t wino_gather_worklist(const int* gq_n, const int* pcnt, int B, int tiles_per_sample, int nrt, unsigned* work, int* total, hipStream_t s);

}  // namespace chk
