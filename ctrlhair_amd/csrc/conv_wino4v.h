// conv_wino4v.h -- Winograd F(4x4,3x3) on the f32 matrix cores with the input transform taken OUT of the contraction kernel: one
// bandwidth-bound pass (wino4v_pack_kernel) writes V = B^T d B of every 6 x 6 patch once, in the exact B-fragment layout of
// v_mfma_f32_16x16x4_f32, and the contraction kernel (wino4v_kernel) is 36 MFMAs + 17 ds_read_b128 per k-step and wave -- no vector
// arithmetic at all between two epilogues.
//
// Why: in conv_wino4.h every (32-row tile, wave pair) re-transforms the patches of its spatial tile -- 144 vector operations per 36
// MFMAs, and a vector instruction costs the SIMD ~3.5 matrix-pipe cycles (vector and matrix pipe do not co-execute on this part:
// DESIGN.md section 7, tools/mfma_coexec2.hip); the same patch is transformed Cout / 16 times (64 x for a 1024-channel ResBlock conv,
// 2 C / 16 times for the SPADE gamma / beta conv of an ACE, whose input -- the 128 hidden channels -- is tiny).  Where the row count
// is large and the level small the extra pass (read X, write 2.25 X) is a few per cent of the conv it serves:
//     pass / conv ~ 65 / rows      (rows = Cout, or 2 C for an ACE)    ->   used for rows >= 512 at <= 64 x 64 pixels (sean_model.cpp).
// Serves architecture.py:82-91 (conv_0 / conv_1 of head / G_middle / up_0), normalization.py:249-257 + :117-153,172-187 (SPADE
// gamma / beta + style convs + the ACE epilogue at 32 / 64 pixels).
//
// Arithmetic: identical to conv_wino4.h -- the pass runs the same wino4_in1d sequence on the same six rows and columns, the contraction
// adds the same products in the same order (channels ascending, four per MFMA), the epilogues are the same functions: results are
// bit-identical to the in-kernel-transform kernels (tests/test_hip_wino.py).
//
// Layout of V in HBM:  V[spatial tile (b, ty, tx)][k-step s][tile group tg 0..3][idx 0..8][lane 0..63][4 floats]
//   spatial tile = 32 x 32 pixels = 8 x 8 tiles of 4 x 4; tile group tg = tile rows 2 tg, 2 tg + 1 (16 tiles, n = 8 (row & 1) + column);
//   lane = 16 kk + n;  float e of idx = V[xi = 4 idx + e][channel 4 s + kk][tile n].   36 864 bytes per (spatial tile, k-step), contiguous:
//   the block's LDS-DMA is a flat copy.
// Stage in LDS = V (36 KB) + fragments idx 0..7 of both 16-row halves of the A image (16 KB) = 53 248 bytes, ring of three = 159 744 of
// the 163 840 bytes (+ a 4 KB sink); the ninth fragment quad of the A image (xi 32..35) does not fit a ring of three and is loaded straight into
// registers (buffer_load_dwordx4, one per lane and k-step, two k-steps ahead like the DMAs).  One barrier per k-step.
#pragma once
#include <type_traits>

#include "conv_wino4.h"

namespace chk {

#ifndef W4V_CARRY
#define W4V_CARRY 1    // 1: group 8 of a k-step runs behind the NEXT k-step's barrier (its operands are registers: covers the LDS latency there)
#endif
#ifndef W4V_ABL
#define W4V_ABL 0      // timing ablations of tools/wino4_bench.hip (wrong results): 1 no barrier, 2 fragment reads of groups 0, 1 only, 4 no DMA, 8 no s_setprio, 16 every k-step from the same addresses, 32 every other DMA piece only
                       // (profiles/r06_wino4v_ablations.txt; bits 64 / 128 / 256 of that table were removed again)
#endif
namespace wino4v {
constexpr int VUN = 2304;                        // V units (16 bytes) per (spatial tile, k-step)
constexpr int VDW = VUN * 4;                     // ... in floats
constexpr int AUN = 1024;                        // A units staged in LDS: idx 0..7 of row half 0, idx 0..7 of row half 1
constexpr int SUN = VUN + AUN;                   // 3328 units per stage
constexpr int NST = 3;
constexpr int LDS_BYTES = NST * SUN * 16 + 4096; // 159 744 + a 4 KB sink for the dummy DMAs of waves 4-7 (below) = all 163 840 bytes
}  // namespace wino4v

// ---- the transform pass ------------------------------------------------------------------------------------------------------------
struct Wino4vPackParams {
    const float* in;        // [B][K][H][pitch], pixel (y, x) of a plane at y * pitch + xoff + x
    float* v;               // V image (layout above), B (H / 32) (W / 32) nks 9216 floats
    int B, K, H, W;         // H % 32 == 0, W % 32 == 0
    int nks;                // k-steps written; channels >= K are zeros
    int pitch, xoff;        // plain activations: pitch = W, xoff = 0; padded hidden-activation planes: wino_apitch(W), WINO_AXOFF
    int padded;             // 1: columns -1 and W are the planes' zero pads (read unchecked); 0: bounds-checked (zeros or reflection)
    int reflect;            // reflection padding (pad 1) instead of zeros
};
// one block = (spatial tile, k-step), wave = tile group, lane = (channel kk, tile n): the lane's 6 x 6 patch -> 36 values -> nine 16-byte
// stores, each a contiguous kilobyte per wave
template <int DUMMY>          // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(256) void wino4v_pack_kernel(const Wino4vPackParams p) {
    const int lane = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int n = lane & 15, kk = lane >> 4;
    const long long blk = blockIdx.x;
    const int s = (int)(blk % p.nks), tile = (int)(blk / p.nks);
    const int ntx = p.W / 32, nty = p.H / 32;
    const int ttx = tile % ntx, tty = (tile / ntx) % nty, b = tile / (ntx * nty);
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const int y0 = tty * 32 + 4 * tyl - 1, x0 = ttx * 32 + 4 * tx;          // x0: first pixel of the tile (the patch starts at x0 - 1)
    const int ch = 4 * s + kk;
    float v[36];
    if (ch < p.K) {
        const float* pl = p.in + ((long long)b * p.K + ch) * p.H * p.pitch + p.xoff;
        int xl = x0 - 1, xr = x0 + 4;
        bool okl = true, okr = true;
        if (!p.padded) {
            if (p.reflect) {
                xl = xl < 0 ? 1 : xl;
                xr = xr >= p.W ? p.W - 2 : xr;
            } else {
                okl = xl >= 0;
                okr = xr < p.W;
            }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            int y = y0 + r;
            if (p.reflect) y = y < 0 ? -y : (y >= p.H ? 2 * p.H - 2 - y : y);
            const bool oky = (unsigned)y < (unsigned)p.H;
            const float* row = pl + (long long)(oky ? y : 0) * p.pitch;
            const f32x4 mid = oky ? *reinterpret_cast<const f32x4*>(row + x0) : (f32x4){0.f, 0.f, 0.f, 0.f};
            const float d0 = (oky && okl) ? row[xl] : 0.f, d5 = (oky && okr) ? row[xr] : 0.f;
            wino4_in1d(d0, mid.x, mid.y, mid.z, mid.w, d5, v[6 * r], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wino4_in1d(v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j], v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j]);
    } else {
#pragma unroll
        for (int x = 0; x < 36; ++x) v[x] = 0.f;
    }
    f32x4* dst = reinterpret_cast<f32x4*>(p.v) + ((blk * 4 + tg) * 9) * 64 + lane;
#pragma unroll
    for (int idx = 0; idx < 9; ++idx) dst[idx * 64] = (f32x4){v[4 * idx], v[4 * idx + 1], v[4 * idx + 2], v[4 * idx + 3]};
}

// ---- the contraction kernel ----------------------------------------------------------------------------------------------------------
// EPI 0: plain conv (Wino4Params, p.v set; p.in unused), EPI 1: SPADE gamma / beta + style images + ACE epilogue (Wino4AceParams, p.v set;
// p.actv unused).  Same block / wave / lane mapping as conv_wino4.h: wave w = row half (w >> 2) x tile group (w & 3), 36 accumulators.
template <int EPI>
__global__ __launch_bounds__(512, 1) void wino4v_kernel(const std::conditional_t<EPI == 0, Wino4Params, Wino4AceParams> p) {
    using namespace wino4v;
    constexpr int ADW = wino4::ADW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int mh = wave >> 2, tg = wave & 3;
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;
    constexpr unsigned SB = SUN * 16u, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    auto task_of = [&](int L, int& rt, int& tile) {        // as conv_wino4.h: 32 consecutive tasks share A images / V tiles through the XCD's L2
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

    // ---- issue side: a flat sequence of k-steps across the block's tasks, two k-steps ahead of the consumers ----------------------------
    // per k-step and thread: V rounds 0..3 (+ round 4: waves 0-3), A rounds 0, 1 (source units tid and 576 + tid: idx 0..7 of each half),
    // and the lane's own ninth quad of its row half straight into a8[parity]
    const unsigned va = (unsigned)tid * 16u, va8 = (unsigned)((9 * mh + 8) * 64 + lane) * 16u;
    int it = lb, is = 0;
    wino_u32x4 d_v, d_a, d_s;
    unsigned so_v = 0, so_a = 0;
    auto issue_task = [&]() {
        int irt, tile;
        task_of(it, irt, tile);
        d_v = wino_rsrc(p.v + (long long)tile * nk * VDW, (unsigned)nk * VDW * 4u);
        if constexpr (EPI == 0) {
            d_a = wino_rsrc(p.wpk + (long long)irt * nk * ADW, (unsigned)nk * ADW * 4u);
        } else {
            d_a = wino_rsrc(p.wpk + (long long)irt * 32 * ADW, 32u * (unsigned)ADW * 4u);
            if (p.wsty) {
                const int ib = tile / (p.ntx * p.nty);
                d_s = wino_rsrc(p.wsty + ((long long)ib * p.nrt + irt) * 6 * ADW, 6u * (unsigned)ADW * 4u);
            }
        }
        so_v = 0;
        so_a = 0;
    };
    issue_task();
    unsigned islot = lds0;
    auto issue_piece = [&](auto pt) {
        constexpr int pc = decltype(pt)::value;
        const unsigned wb = islot + (unsigned)wave * 1024u;
        if constexpr (W4V_ABL & 4) return;
        if constexpr (W4V_ABL & 32)
            if (pc & 1) return;
        if constexpr (pc < 4) wino_dma16(va, d_v, so_v + (unsigned)pc * 8192u, wb + (unsigned)pc * 8192u);
        else wino_dma16(va, d_a, so_a + (unsigned)(pc - 4) * 9216u, wb + VUN * 16u + (unsigned)(pc - 4) * 8192u);
    };
    auto issue_direct = [&](f32x4& dst) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(va8), "s"(d_a), "s"(so_a) : "memory");
    };
    // W4V_CARRY: the ninth quad is loaded ONE k-step ahead (not two), from the A image the previous issue step addressed
    wino_u32x4 d_ap;
    unsigned so_ap = 0;
    auto issue_direct_prev = [&](f32x4& dst) {
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(va8), "s"(d_ap), "s"(so_ap) : "memory");
    };
    auto issue_tail = [&]() {
        // V round 4 is half a round (waves 0-3); waves 4-7 issue a DMA too, from beyond num_records (zeros) into the sink behind the ring,
        // so that every wave has the same number of loads in flight and ONE counted wait serves all (no branch around the wait)
        if constexpr (!(W4V_ABL & 4)) {
            if (wave < 4) wino_dma16(va, d_v, so_v + 4u * 8192u, islot + (unsigned)wave * 1024u + 4u * 8192u);
            else wino_dma16(0x80000000u, d_v, 0u, lds0 + RING + (unsigned)(wave - 4) * 1024u);
        }
        islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
        if constexpr (W4V_CARRY) {
            d_ap = d_a;
            so_ap = so_a;
        }
        if constexpr (!(W4V_ABL & 16)) {
            so_v += (unsigned)VDW * 4u;
            so_a += (unsigned)ADW * 4u;
        }
        ++is;
        if constexpr (EPI == 1) {
            if (is == 32 && nk > 32) {         // the style images of the task's sample follow the hidden channels
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
                so_v -= (unsigned)VDW * 4u;
                so_a -= (unsigned)ADW * 4u;
            }
        }
    };
    // at the top of k-step q the eight loads of k-step q + 1 may still be in flight.  The ninth quad of k-step q was the output of an asm
    // statement long past; the compiler may copy or use it from that point on, so the value is re-defined by an (empty) asm BEHIND the
    // wait and a scheduling fence: whatever register shuffling the tied operand causes happens after the data has landed
    auto wait_ring = [&](f32x4& a8q) {
        if constexpr (W4V_ABL & 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a8q));
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------------------------
    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a8[2];
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const unsigned a_off = (unsigned)(VUN + mh * 512 + lane) * 4u, v_off = (unsigned)(tg * 9 * 64 + lane) * 4u;      // floats
    auto stage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };

    f32x4 v8c = {0.f, 0.f, 0.f, 0.f};         // W4V_CARRY: V fragments idx 8 of the previous k-step (zero: nothing pending)
    {   // prologue: k-steps 0 and 1
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{}); issue_piece(WInt<4>{}); issue_piece(WInt<5>{});
        issue_direct(a8[0]);
        issue_tail();
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{}); issue_piece(WInt<4>{}); issue_piece(WInt<5>{});
        if constexpr (W4V_CARRY) a8[1] = (f32x4){0.f, 0.f, 0.f, 0.f};       // (k-step 0 loads the quad of k-step 1 itself)
        else issue_direct(a8[1]);
        issue_tail();
        // W4V_CARRY: the second k-step of the prologue is seven loads, not eight -- the counted wait at the top of k-step 0 would let the LAST
        // load of stage 0 (V round 4: fragments 5..8 of tile group 3) stay in flight while the k-step reads it.  Once per block: wait for it here.
        // (Found as a one-in-ten-runs difference between this route and the in-kernel transform; both k-steps counted eight before the carry.)
        if constexpr (W4V_CARRY) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    unsigned rslot = lds0;
    auto mfma4 = [&](const f32x4& a, const f32x4& b, int g) {
        acc[4 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[4 * g], 0, 0, 0);
        acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[4 * g + 1], 0, 0, 0);
        acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[4 * g + 2], 0, 0, 0);
        acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[4 * g + 3], 0, 0, 0);
    };
    // One k-step q.  Fragment reads two groups ahead; the DMAs of k-step q + 2 go out one piece per group.
    //   W4V_CARRY = 1: behind the barrier the reads of groups 0, 1 are issued and then the four MFMAs of group 8 of k-step q - 1 run -- their
    //     operands (the ninth A quad of q - 1, loaded during q - 2, and V idx 8 of stage q - 1, read before the barrier) are registers, so the
    //     matrix pipe works while the LDS answers; the quad's register is then reloaded for k-step q + 1.  A task's last group 8 is
    //     flushed before its epilogue.
    //   W4V_CARRY = 0: group 8 first in its own k-step (its A quad is a register: free for the load of k-step q + 2 right after).
    auto kstep = [&](auto par) {
        constexpr int PAR = decltype(par)::value;
        constexpr int QR = W4V_CARRY ? (PAR ^ 1) : PAR;       // the quad register consumed (and reloaded) in this k-step
        wait_ring(a8[QR]);
        if constexpr (!(W4V_ABL & 1)) __syncthreads();
        const float* sp = stage(rslot);
        const f32x4* ap = reinterpret_cast<const f32x4*>(sp + a_off);
        const f32x4* vp = reinterpret_cast<const f32x4*>(sp + v_off);
        f32x4 A[3], V[3];
        f32x4 v8 = v8c;
        if constexpr (!W4V_CARRY) v8 = vp[8 * 64];
        A[0] = ap[0];
        V[0] = vp[0];
        A[1] = ap[64];
        V[1] = vp[64];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(W4V_ABL & 8)) __builtin_amdgcn_s_setprio(1);
        mfma4(a8[QR], v8, 8);
        if constexpr (!(W4V_ABL & 8)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (W4V_CARRY) issue_direct_prev(a8[QR]);
        else issue_direct(a8[QR]);
        __builtin_amdgcn_sched_barrier(0);
        auto group = [&](auto gt) {
            constexpr int g = decltype(gt)::value;
            if constexpr (g + 2 < 8 && !(W4V_ABL & 2)) {
                A[(g + 2) % 3] = ap[(g + 2) * 64];
                V[(g + 2) % 3] = vp[(g + 2) * 64];
            }
            if constexpr (W4V_CARRY && g == 6) v8c = vp[8 * 64];       // (read before the next barrier: the stage is overwritten after it)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(W4V_ABL & 8)) __builtin_amdgcn_s_setprio(1);
            mfma4(A[(W4V_ABL & 2) ? (g & 1) : g % 3], V[(W4V_ABL & 2) ? (g & 1) : g % 3], g);
            if constexpr (!(W4V_ABL & 8)) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g < 6) issue_piece(WInt<g>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
        group(WInt<6>{}); group(WInt<7>{});
        issue_tail();
        rslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
    };

    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        for (int cs = 0; cs < nk; cs += 2) {      // (nk is even: the launchers)
            kstep(WInt<0>{});
            kstep(WInt<1>{});
        }
        if constexpr (W4V_CARRY) {                // group 8 of the task's last k-step (quad a8[1], loaded during the k-step before it)
            wait_ring(a8[1]);
            mfma4(a8[1], v8c, 8);
            v8c = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int crt, tile;
        task_of(ct, crt, tile);
        if constexpr (EPI == 0) wino4_plain_epilogue(p, acc, crt, tile, mh, kk, tyl, tx);
        else wino4_ace_epilogue(p, acc, crt, tile, mh, kk, tyl, tx);
#pragma unroll
        for (int x2 = 0; x2 < 36; ++x2) acc[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the epilogue's loads / stores share the counter with the ring: drain once per task (the two quads in flight arrive with it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(a8[0]), "+v"(a8[1]));
    }
}

// bytes of the V image of a conv input / of an ACE's hidden activations (K channels in nks = ceil(K / 4) k-steps, rounded up to even)
inline size_t wino4v_bytes(int B, int H, int W, int nks) { return (size_t)B * (H / 32) * (W / 32) * nks * wino4v::VDW * sizeof(float); }
// Does the extra pass pay?  pass / conv ~ 65 / rows (header); the contraction kernel gains ~20 %.
inline bool wino4v_pays(int rows, int r) { return rows >= 512 && r <= 64; }
hipError_t wino4v_pack(const Wino4vPackParams& p, hipStream_t s);       // conv_inst_wino4.hip
hipError_t conv_wino4v_plain(Wino4Params p, hipStream_t s);             // p.v = V image of p's input (wino4v_pack), p.in unused
hipError_t conv_wino4v_ace(Wino4AceParams p, hipStream_t s);            // p.v = V image of the hidden activations (+ one-hot planes)

}  // namespace chk
