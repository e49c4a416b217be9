// tools/wino4s_experiment.h -- EXPERIMENT, not part of the library (tools/wino4_bench.hip -DW4S_EXPERIMENT): the F(4x4,3x3) conv of
// conv_wino4.h with the input transform done ONCE per pair of waves (round 6).  Correct (bit-identical to wino4_plain_kernel on every
// shape of the tool) and NOT faster: 16.54 vs 16.42 ms over the twelve ResBlock convs of a step.  The ablations (-DW4S_ABL=...,
// profiles/r06_wino4s_ablation.txt) say why: with everything but the MFMAs, the first barrier and the epilogue removed the kernel runs
// at 129 TFLOP/s executed (up_1 conv_0: 1.20 ms against 1.96); the producer's transform arithmetic alone is worth 17 % although only one
// wave of a SIMD's two carries it -- vector and matrix instructions of DIFFERENT waves of a SIMD do not overlap either -- the fragment
// exchange 9 %, the DMAs 5 %, A-fragment reads and the second barrier 1 % each.  Halving the arithmetic per SIMD buys what the
// exchange costs.  Kept as the record of that measurement.
//
// In wino4_plain_kernel the two waves that share a group of 16 tiles (one per 16-row half of the task's 32 GEMM rows) each transform the
// same 6 x 6 patches: 144 vector operations + 18 LDS reads per wave and k-step next to 36 MFMAs, and the matrix pipe waits while the
// vector unit of its SIMD works (58 % busy in profiles/r06_f32_pmc.md).  Here wave w < 4 (the PRODUCER of tile group w) transforms and
// hands the 36 B fragments of its lanes to wave w + 4 (the CONSUMER, same SIMD, same lane -> (tile, channel) mapping) through LDS:
// [tile group][lane][36 floats], nine 16-byte writes / reads per lane, conflict-free (144-byte lane stride).  The consumer's k-step is
// 36 MFMAs + 9 A-fragment reads + 9 B-fragment reads and no vector arithmetic.  Same operations on the same values in the same order as
// wino4_plain_kernel: the results are bit-identical (checked by the tool).
//
// LDS (all 160 KB): the exchange buffer takes 36 KB, so the ring of four 40 KB stages becomes two rings with the depths their
// consumers need -- patches FOUR deep (22.5 KB each; first touches of HBM, read by the transform one k-step before the MFMAs that use
// the result: issued three k-steps ahead), A images TWO deep (18 KB each; L2 hits, issued one k-step ahead, FIRST in the k-step):
//     [0, 90 112) patches | [90 112, 126 976) A images | [126 976, 163 840) B-fragment exchange
// Counted waits: within a k-step every wave issues its A DMAs before its patch DMAs, so `s_waitcnt vmcnt(#patch DMAs of a k-step)` at
// the top of a k-step leaves exactly the newest patch in flight.  Two barriers per k-step: B1 (top: DMAs landed, fragments written) and
// B2 (after group 4: the consumers have read the fragments -- the producers overwrite them after group 8).
// Producer and consumer are two straight-line code paths (a branch inside the k-step costs far more than its instructions,
// conv_wino.h); both execute the same sequence of barriers.
#pragma once
#include "../ctrlhair_amd/csrc/conv_wino4.h"

#ifndef W4S_ABL
#define W4S_ABL 0      // timing ablations of tools/wino4_bench.hip (wrong results): 1 no transform arithmetic, 2 no patch reads, 4 no fragment exchange,
#endif                 // 8 no second barrier, 16 no A-fragment reads, 32 no DMAs

namespace chk {

namespace wino4s {
constexpr int NPS = 4, NAS = 2;
constexpr unsigned PB = wino4::PSLOTS * 16u;      // 22 528
constexpr unsigned AB = wino4::AUNITS * 16u;      // 18 432
constexpr unsigned A0 = NPS * PB;                 // 90 112
constexpr unsigned X0 = A0 + NAS * AB;            // 126 976
constexpr unsigned XB = 64u * 36u * 4u;           // 9 216 per tile group
constexpr int LDS_BYTES = (int)(X0 + 4u * XB);    // 163 840
static_assert(LDS_BYTES == 163840, "the three regions fill the CU's LDS exactly");
}  // namespace wino4s

template <int MODE>
__global__ __launch_bounds__(512, 1) void wino4s_plain_kernel(const Wino4Params p) {
    using namespace wino4;
    constexpr bool REFL = (MODE & 1) != 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kk = lane >> 4;
    const int mh = wave >> 2, tg = wave & 3;       // row half (0: producer, 1: consumer), tile group
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;
    const int HW = p.H * p.W;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    auto task_of = [&](int L, int& rt, int& tile) {      // as wino4_plain_kernel
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

    // ---- issue side: two cursors over the block's flat sequence of (task, k-step) ------------------------------------------------
    unsigned voff[3];
    const unsigned va = (unsigned)tid * 16u;
    const unsigned wb = (unsigned)wave * 1024u;
    int itP = lb, isP = 0, itA = lb, isA = 0;
    wino_u32x4 d_in, d_a;
    unsigned so_in = 0, so_a = 0;
    unsigned pslot = lds0, aslot = lds0 + wino4s::A0;
    auto task_P = [&]() {
        int irt, tile;
        task_of(itP, irt, tile);
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
        if (wave >= 6) voff[2] = voff[1];          // (waves 6-7 have no third round: they repeat their second, see issue_P)
        d_in = wino_rsrc(p.in + (long long)ib * p.Cin * HW, (unsigned)p.Cin * HW * 4u);
        so_in = 0;
    };
    auto task_A = [&]() {
        int irt, tile;
        task_of(itA, irt, tile);
        d_a = wino_rsrc(p.wpk + (long long)irt * p.nks * ADW, (unsigned)p.nks * ADW * 4u);
        so_a = 0;
    };
    // Three A and three patch DMAs per wave and k-step, no branch around any of them: the third round of the A image is 128 units (waves
    // 0-1) and the third patch round 384 (waves 0-5) -- the other waves repeat their second round (same source, same destination).
    const unsigned a3 = wave < 2 ? 2u * 8192u : 8192u, p3 = wave < 6 ? 2u * 8192u : 8192u;
    auto issue_A = [&](auto pt) {
        constexpr int pc = decltype(pt)::value;
        const unsigned o = pc < 2 ? (unsigned)pc * 8192u : a3;
        if constexpr (!(W4S_ABL & 32)) wino_dma16(va, d_a, so_a + o, aslot + wb + o);
    };
    auto issue_P = [&](auto pt) {
        constexpr int pc = decltype(pt)::value;
        if constexpr (W4S_ABL & 32) return;
        if constexpr (pc < 2) wino_dma16(voff[pc], d_in, so_in, pslot + wb + (unsigned)pc * 8192u);
        else wino_dma16(voff[2], d_in, so_in, pslot + wb + p3);
    };
    auto advance_A = [&]() {
        aslot = aslot == lds0 + wino4s::A0 ? aslot + wino4s::AB : lds0 + wino4s::A0;
        so_a += (unsigned)ADW * 4u;
        if (++isA == nk) {
            if (itA + G < p.ntasks) {
                itA += G;
                isA = 0;
                task_A();
            } else {                   // past the end: keep re-issuing the last k-step (never read; keeps the vmcnt counting uniform)
                isA = nk - 1;
                so_a -= (unsigned)ADW * 4u;
            }
        }
    };
    auto advance_P = [&]() {
        pslot = pslot + wino4s::PB == lds0 + wino4s::A0 ? lds0 : pslot + wino4s::PB;
        so_in += 16u * (unsigned)HW;
        if (++isP == nk) {
            if (itP + G < p.ntasks) {
                itP += G;
                isP = 0;
                task_P();
            } else {
                isP = nk - 1;
                so_in -= 16u * (unsigned)HW;
            }
        }
    };
    // the newest patch (three DMAs of this wave) may still be in flight at the top of a k-step
    auto wait_ring = [&]() {
        if constexpr (!(W4S_ABL & 32)) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    };

    // ---- consumer side ---------------------------------------------------------------------------------------------------
    f32x4 acc[36];
#pragma unroll
    for (int x = 0; x < 36; ++x) acc[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tx = n & 7, tyl = 2 * tg + (n >> 3);
    const int boff = kk * (PPL * 4) + (4 * tyl) * (PUN * 4) + 4 * tx + 3;      // this lane's patch origin (floats) inside a patch slot
    auto pstage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem) + (slot - lds0) / 4) + 9 * mh * 64 + lane; };
    f32x4* xq = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(smem) + wino4s::X0 + (unsigned)tg * wino4s::XB + (unsigned)lane * 144u);

    task_P();
    task_A();
    for (int i = 0; i < 3; ++i) {
        issue_P(WInt<0>{}); issue_P(WInt<1>{}); issue_P(WInt<2>{});
        advance_P();
    }
    issue_A(WInt<0>{}); issue_A(WInt<1>{}); issue_A(WInt<2>{});
    advance_A();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned rp = lds0, ra = lds0 + wino4s::A0;        // patch slot of the k-step being multiplied, its A slot
    auto next_p = [&](unsigned s) { return s + wino4s::PB == lds0 + wino4s::A0 ? lds0 : s + wino4s::PB; };
    auto next_a = [&](unsigned s) { return s == lds0 + wino4s::A0 ? s + wino4s::AB : lds0 + wino4s::A0; };
    // the MFMAs of group g on the B fragments vc + the issue pieces that ride behind the groups (both paths)
    auto mfma_group = [&](auto gt, const f32x4 c, float (&vc)[36]) {
        constexpr int g = decltype(gt)::value;
        __builtin_amdgcn_s_setprio(1);
        acc[4 * g] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x, vc[4 * g], acc[4 * g], 0, 0, 0);
        acc[4 * g + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.y, vc[4 * g + 1], acc[4 * g + 1], 0, 0, 0);
        acc[4 * g + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.z, vc[4 * g + 2], acc[4 * g + 2], 0, 0, 0);
        acc[4 * g + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.w, vc[4 * g + 3], acc[4 * g + 3], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto issue_behind = [&](auto gt) {
        constexpr int g = decltype(gt)::value;
        if constexpr (g < 3) issue_A(gt);                     // (all A DMAs of a k-step before its patch DMAs: wait_ring)
        if constexpr (g >= 3 && g < 6) issue_P(WInt<g - 3>{});
        if constexpr (g == 8) {                               // the only branches of the issue side, once per k-step behind the last group
            advance_A();
            advance_P();
        }
    };
    auto epilogue = [&](int ct) {
        int crt, tile;
        task_of(ct, crt, tile);
        wino4_plain_epilogue(p, acc, crt, tile, mh, kk, tyl, tx);
#pragma unroll
        for (int x2 = 0; x2 < 36; ++x2) acc[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's loads / stores share the counter with the rings: drain once per task
    };

    if (mh == 0) {
        // ================= producer: transforms the NEXT k-step's patches behind its MFMA groups, publishes them after group 8 =========
        bool eL = false, eR = false;
        auto edge_of = [&](int L, bool& l, bool& r) {
            int rt_, tile_;
            task_of(L < p.ntasks ? L : p.ntasks - 1, rt_, tile_);
            const int ttx_ = tile_ % p.ntx;
            l = ttx_ == 0 && tx == 0;
            r = ttx_ == p.ntx - 1 && tx == 7;
        };
        auto load_row = [&](const float* sp, int r, float (&d)[6]) {
            const float* q = sp + boff + r * (PUN * 4);
            const f32x4 mid = *reinterpret_cast<const f32x4*>(q + 1);
            d[1] = mid.x; d[2] = mid.y; d[3] = mid.z; d[4] = mid.w;
            d[0] = q[0];
            d[5] = q[5];
            if constexpr (REFL) {
                d[0] = eL ? d[2] : d[0];
                d[5] = eR ? d[3] : d[5];
            }
        };
        auto publish = [&](float (&vx)[36]) {
#pragma unroll
            for (int i = 0; i < 9; ++i) xq[i] = (f32x4){vx[4 * i], vx[4 * i + 1], vx[4 * i + 2], vx[4 * i + 3]};
        };
        float v[36], w[36];
        if constexpr (REFL) edge_of(lb, eL, eR);
        {   // B fragments of the first k-step
            const float* sp = pstage(lds0);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float d[6];
                load_row(sp, r, d);
                wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], v[6 * r], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5]);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                wino4_in1d(v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j], v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j]);
            publish(v);
        }
        auto kstep = [&](float (&vc)[36], float (&vx)[36]) {
            wait_ring();
            __syncthreads();                                   // B1
            const unsigned np = next_p(rp);
            const f32x4* ap = a_ptr(ra);
            const float* spn = pstage(np);
            f32x4 F[2];
            F[0] = ap[0];
            float d[6];
            auto group = [&](auto gt) {
                constexpr int g = decltype(gt)::value;
                if constexpr (g + 1 < 9 && !(W4S_ABL & 16)) F[(g + 1) & 1] = ap[(g + 1) * 64];
                if constexpr (g < 6 && !(W4S_ABL & 2)) load_row(spn, g, d);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(gt, F[g & 1], vc);
                if constexpr (g < 6 && !(W4S_ABL & 1))
                    wino4_in1d(d[0], d[1], d[2], d[3], d[4], d[5], vx[6 * g], vx[6 * g + 1], vx[6 * g + 2], vx[6 * g + 3], vx[6 * g + 4], vx[6 * g + 5]);
                if constexpr (g >= 6 && !(W4S_ABL & 1)) {
#pragma unroll
                    for (int j = 2 * (g - 6); j < 2 * (g - 6) + 2; ++j)
                        wino4_in1d(vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j], vx[30 + j], vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j],
                                   vx[30 + j]);
                }
                __builtin_amdgcn_sched_barrier(0);
                issue_behind(gt);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g == 4 && !(W4S_ABL & 8)) {      // B2: the consumers hold the fragments of this k-step
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
            group(WInt<6>{}); group(WInt<7>{}); group(WInt<8>{});
            if constexpr (!(W4S_ABL & 4)) publish(vx);
            rp = np;
            ra = next_a(ra);
        };
        for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
            for (int cs = 0; cs < nk; cs += 2) {
                kstep(v, w);          // (nks is even: the launcher)
                if constexpr (REFL)
                    if (cs + 2 >= nk) edge_of(ct + G, eL, eR);
                kstep(w, v);
            }
            epilogue(ct);
        }
    } else {
        // ================= consumer: B fragments from the exchange buffer, no vector arithmetic ===================================
        float vc[36];
        if constexpr ((W4S_ABL & 4) != 0)
            for (int i = 0; i < 36; ++i) vc[i] = 1.f + (float)lane;
        auto kstep = [&]() {
            wait_ring();
            __syncthreads();                                   // B1
            const f32x4* ap = a_ptr(ra);
            f32x4 F[2];
            F[0] = ap[0];
#pragma unroll
            for (int i = 0; i < ((W4S_ABL & 4) ? 0 : 9); ++i) {
                const f32x4 q = xq[i];
                vc[4 * i] = q.x; vc[4 * i + 1] = q.y; vc[4 * i + 2] = q.z; vc[4 * i + 3] = q.w;
            }
            auto group = [&](auto gt) {
                constexpr int g = decltype(gt)::value;
                if constexpr (g + 1 < 9 && !(W4S_ABL & 16)) F[(g + 1) & 1] = ap[(g + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(gt, F[g & 1], vc);
                __builtin_amdgcn_sched_barrier(0);
                issue_behind(gt);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g == 4 && !(W4S_ABL & 8)) {      // B2 (every fragment read has landed: group 4 is behind them all)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            group(WInt<0>{}); group(WInt<1>{}); group(WInt<2>{}); group(WInt<3>{}); group(WInt<4>{}); group(WInt<5>{});
            group(WInt<6>{}); group(WInt<7>{}); group(WInt<8>{});
            ra = next_a(ra);
        };
        for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
            for (int cs = 0; cs < nk; cs += 2) {
                kstep();
                kstep();
            }
            epilogue(ct);
        }
    }
}

}  // namespace chk
