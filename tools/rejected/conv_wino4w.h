// tools/rejected/conv_wino4w.h -- NOT part of the library (nothing includes it; it needs a `wide` switch in Wino4Params and in the launcher,
// see the round-5 history): a CORRECT F(4x4,3x3) kernel with a wave tile of 32 rows on one wave per SIMD -- 72 accumulators, 64 of them
// pinned to the accumulator half of the register file through inline-assembly MFMAs -- that runs at the speed of the shipped kernel
// (sum over the ResBlock convs of a step: 15.05 vs 14.50 ms; tools/wino4_bench.hip, same box, all shapes OK against the double-precision
// direct conv).  Half the transform work and half the LDS reads per MFMA bought nothing: whatever holds the F(4x4) k-step at ~58 % of the
// matrix pipe, it is not the vector-instruction count.
//
// conv_wino4w.h -- the F(4x4,3x3) conv of conv_wino4.h on ONE wave per SIMD with a wave tile of 32 GEMM rows (round 5).
//
// Why: conv_wino4.h's waves own 16 rows x 16 blocks (36 accumulators of 16 x 16 = 144 registers, two waves per SIMD) and each
// of them transforms its blocks' patches -- 4 vector operations per MFMA, which with the LDS reads, the DMA issue and the scalar
// bookkeeping makes ~5.9 instructions beside every MFMA where the matrix pipe hides about five: the
// k-step is bound by the SIMD's instruction issue and the matrix pipe is busy 58 % of the time.  Here a wave owns BOTH 16-row halves
// of its blocks: 72 accumulators (288 registers), one transform per 72 MFMAs = 2 vector operations per MFMA.  That needs the whole
// register file of the SIMD (512 registers per lane: one 256-thread block per CU), and it needs the accumulators placed BY HAND:
// hipcc puts every MFMA result of a function either in the accumulator half (AGPR) or in the vector half of the file, and with more
// than 256 accumulator registers it copies accumulators between the halves around every MFMA.  The MFMAs are inline assembly: 64
// accumulators carry the constraint "a" (AGPR), 8 the constraint "v".  hipcc does not model an assembly statement: the waits on the
// LDS reads feeding one are still placed by the compiler (the operands are its own registers), the MFMA-result -> reader hazard is
// covered by hand (12 states behind the last MFMA of a task, before the epilogue reads the accumulators).
//
// One wave per SIMD has no partner to hide anything behind: the instruction stream itself is interleaved -- every MFMA is followed by
// at most ONE chunk of three transform operations and at most one LDS read or DMA (the 72 issue slots of a k-step are filled from a
// compile-time schedule, W4wSlot below); A fragments and patch rows are read one group (eight MFMAs) ahead of their use.
//
// Everything else -- task order, stage layout, DMA ring as one flat sequence across tasks, weight images, epilogue arithmetic -- is
// conv_wino4.h's; the two kernels produce bit-identical results (same operations per output in the same order).
#pragma once
#include "../../ctrlhair_amd/csrc/conv_wino4.h"

namespace chk {

#ifdef W4W_STAMP
__device__ unsigned long long w4w_stamps[4 * 64 * 12];      // [wave][k-step][stamp]: tools/w4w_timeline.hip
#endif

template <bool AG>
__device__ __forceinline__ void w4w_mfma(f32x4& acc, float a, float b) {
    if constexpr (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// the one-dimensional input transform of conv_wino4.h (wino4_in1d) in four chunks of three operations
struct W4wIn1d {
    float d0, d1, d2, d3, d4, d5, a, b, c, t, u, w;
};
// (the empty statements pin a chunk behind the MFMA of its slot: without them the instruction selector sinks the whole transform -- pure
//  operations whose results are only needed in the next k-step -- to the end of the k-step, where nothing hides it)
template <int CH>
__device__ __forceinline__ void w4w_chunk(W4wIn1d& s, float& o0, float& o1, float& o2, float& o3, float& o4, float& o5) {
    if constexpr (CH == 0) {
        s.a = __builtin_fmaf(-4.f, s.d2, s.d4);
        s.b = __builtin_fmaf(-4.f, s.d1, s.d3);
        s.c = s.d4 - s.d2;
        asm volatile("" : "+v"(s.a), "+v"(s.b), "+v"(s.c));
    } else if constexpr (CH == 1) {
        s.t = s.d3 - s.d1;
        s.u = __builtin_fmaf(-5.f, s.d2, s.d4);
        s.w = __builtin_fmaf(-5.f, s.d3, s.d5);
        asm volatile("" : "+v"(s.t), "+v"(s.u), "+v"(s.w));
    } else if constexpr (CH == 2) {
        const float r0 = __builtin_fmaf(4.f, s.d0, s.u), r5 = __builtin_fmaf(4.f, s.d1, s.w), r1 = s.a + s.b;
        o0 = r0; o5 = r5; o1 = r1;
        asm volatile("" : "+v"(o0), "+v"(o5), "+v"(o1));
    } else {
        const float r2 = s.a - s.b, r3 = __builtin_fmaf(2.f, s.t, s.c), r4 = __builtin_fmaf(-2.f, s.t, s.c);
        o2 = r2; o3 = r3; o4 = r4;
        asm volatile("" : "+v"(o2), "+v"(o3), "+v"(o4));
    }
}

template <int I, int N, class F>
__device__ __forceinline__ void w4w_for(F&& f) {
    if constexpr (I < N) {
        f(WInt<I>{});
        w4w_for<I + 1, N>(f);
    }
}

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void wino4_plain_w_kernel(const Wino4Params p) {
    using namespace wino4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = tile group: tile rows 2 wave, 2 wave + 1
    const int n = lane & 15, kk = lane >> 4;
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nk = p.nks;
    const int HW = p.H * p.W;
    constexpr unsigned SB = SUNITS * 16u, RING = NST * SB;
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    auto task_of = [&](int L, int& rt, int& tile) {      // conv_wino4.h
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

    // ---- issue side: a stage = 1408 patch slots (1360 units) + 1152 A units of 16 bytes; 256 threads: patch rounds 0-4 (+ round 5 on
    //      waves 0-1: units 1280-1407), A rounds 0-3 (+ round 4 on waves 0-1) ------------------------------------------------
    unsigned voff[6];
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
        for (int i = 0; i < 6; ++i) {
            const int u = i * 256 + tid;
            const int k4 = u / PPL, rem = u - k4 * PPL;
            const int py = rem / PUN, ux = rem - py * PUN;
            const int y = y0 + py, x = x0 + 4 * ux;
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
    // pieces 0-4: patch rounds, 5-8: A rounds -- one per group of eight MFMAs; issue_tail: the partial rounds + advance
    auto issue_piece = [&](auto pt) {
        constexpr int pc = decltype(pt)::value;
        const unsigned wb = islot + (unsigned)wave * 1024u;
        if constexpr (pc < 5) wino_dma16(voff[pc], d_in, so_in, wb + (unsigned)pc * 4096u);
        else wino_dma16(va, d_a, so_a + (unsigned)(pc - 5) * 4096u, wb + PSLOTS * 16u + (unsigned)(pc - 5) * 4096u);
    };
    auto issue_tail = [&]() {
        const unsigned wb = islot + (unsigned)wave * 1024u;
        if (wave < 2) {
            wino_dma16(voff[5], d_in, so_in, wb + 5u * 4096u);
            wino_dma16(va, d_a, so_a + 4u * 4096u, wb + PSLOTS * 16u + 4u * 4096u);
        }
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
        issue_piece(WInt<0>{}); issue_piece(WInt<1>{}); issue_piece(WInt<2>{}); issue_piece(WInt<3>{}); issue_piece(WInt<4>{});
        issue_piece(WInt<5>{}); issue_piece(WInt<6>{}); issue_piece(WInt<7>{}); issue_piece(WInt<8>{});
        issue_tail();
    };
    // one k-step's DMAs of this wave (11 / 9) may still be in flight at the top of a k-step: the two stages it reads were issued before
    auto wait_ring = [&]() {
        if (wave < 2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    };

    // ---- consumer side: accumulator (half h, position xi) = index 36 h + xi; 0-63 in the accumulator half of the file, 64-71 in the
    //      vector half ---------------------------------------------------------------------------------------------------------
    f32x4 accA[64], accV[8];
#pragma unroll
    for (int x = 0; x < 64; ++x) accA[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < 8; ++x) accV[x] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tx = n & 7, tyl = 2 * wave + (n >> 3);
    const int boff = kk * (PPL * 4) + (4 * tyl) * (PUN * 4) + 4 * tx + 3;      // this lane's patch origin (floats) inside a stage
    auto stage = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    struct Row { f32x4 mid; float e0, e5; };
    auto load_row = [&](const float* sp, int r, Row& d) {                       // patch row r of the lane's block: 1 + 4 + 1 floats
        const float* q = sp + boff + r * (PUN * 4);
        d.e0 = q[0];
        d.mid = *reinterpret_cast<const f32x4*>(q + 1);
        d.e5 = q[5];
    };
    auto a_ptr = [&](unsigned slot) { return reinterpret_cast<const f32x4*>(stage(slot) + PSLOTS * 4) + lane; };

    issue_kstep();
    issue_kstep();
    issue_kstep();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float v[36], w[36];
    {   // B fragments of the first k-step
        const float* sp = stage(lds0);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            Row d;
            load_row(sp, r, d);
            wino4_in1d(d.e0, d.mid.x, d.mid.y, d.mid.z, d.mid.w, d.e5, v[6 * r], v[6 * r + 1], v[6 * r + 2], v[6 * r + 3], v[6 * r + 4], v[6 * r + 5]);
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            wino4_in1d(v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j], v[j], v[6 + j], v[12 + j], v[18 + j], v[24 + j], v[30 + j]);
    }
#ifdef W4W_STAMP
    int kcnt = 0;
#endif
    unsigned rslot = lds0;
    // A fragments of group 0 of the first k-step (later ones are read during the previous k-step's last group)
    f32x4 F[2][2];      // [ring][half]
    {
        const f32x4* ap = a_ptr(lds0);
        F[0][0] = ap[0];
        F[0][1] = ap[9 * 64];
    }

    // one k-step = 72 MFMAs = 72 issue slots; slot S = 8 g + s (group g: positions 4 g .. 4 g + 3, s = 2 e + h: position 4 g + e, half h).
    // Beside its MFMA a slot carries at most one chunk of the NEXT k-step's transform and one memory instruction:
    //   rows:     row r's four chunks in slots 8 r + 4 .. 8 r + 7 (its LDS reads: slot 8 r - 3, i.e. one group earlier; row 0: top of the k-step)
    //   columns:  column j's four chunks in slots 48 + 4 j .. 48 + 4 j + 3
    //   A reads:  group g + 1's two fragments in slots 8 g + 1, 8 g + 2 (group 8: the NEXT k-step's group 0, from the next stage)
    //   DMA:      piece g in slot 8 g + 3
    auto kstep = [&](auto Pt, float (&vc)[36], float (&vx)[36]) {
        constexpr int P = decltype(Pt)::value;          // parity of the A ring at group 0 (nine groups per k-step: it alternates)
#ifdef W4W_STAMP
        unsigned long long ts[11];
        asm volatile("s_memtime %0" : "=s"(ts[0]));
#endif
        wait_ring();
        __syncthreads();
#ifdef W4W_STAMP
        asm volatile("s_memtime %0" : "=s"(ts[1]));
#endif
        const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
        const f32x4* ap = a_ptr(rslot);
        const f32x4* apn = a_ptr(nslot);
        const float* spn = stage(nslot);               // (k-step q + 1 was verified together with q)
        Row dr[2];
        load_row(spn, 0, dr[0]);
        W4wIn1d st;
        auto slot = [&](auto St) {
            constexpr int S = decltype(St)::value, g = S / 8, s = S % 8, e = s >> 1, h = s & 1, xi = 4 * g + e, ai = 36 * h + xi;
            const float af = F[(g + P) & 1][h][e];
            if constexpr (ai < 64) w4w_mfma<true>(accA[ai], af, vc[xi]);
            else w4w_mfma<false>(accV[ai - 64], af, vc[xi]);
            // ---- the chunk of this slot
            if constexpr (S < 48 && s >= 4) {
                constexpr int r = g, ch = s - 4;
                if constexpr (ch == 0) {
                    const Row& d = dr[r & 1];
                    st.d0 = d.e0; st.d1 = d.mid.x; st.d2 = d.mid.y; st.d3 = d.mid.z; st.d4 = d.mid.w; st.d5 = d.e5;
                }
                w4w_chunk<ch>(st, vx[6 * r], vx[6 * r + 1], vx[6 * r + 2], vx[6 * r + 3], vx[6 * r + 4], vx[6 * r + 5]);
            }
            if constexpr (S >= 48) {
                constexpr int j = (S - 48) / 4, ch = (S - 48) % 4;
                if constexpr (ch == 0) {
                    st.d0 = vx[j]; st.d1 = vx[6 + j]; st.d2 = vx[12 + j]; st.d3 = vx[18 + j]; st.d4 = vx[24 + j]; st.d5 = vx[30 + j];
                }
                w4w_chunk<ch>(st, vx[j], vx[6 + j], vx[12 + j], vx[18 + j], vx[24 + j], vx[30 + j]);
            }
            // ---- the memory instruction of this slot
            if constexpr (s == 1 || s == 2) {
                constexpr int hh = s - 1;
                if constexpr (g < 8) F[(g + 1 + P) & 1][hh] = ap[(9 * hh + g + 1) * 64];
                else F[(g + 1 + P) & 1][hh] = apn[(9 * hh) * 64];      // group 0 of the next k-step (parity 1 - P), from the next stage
            }
            if constexpr (s == 5 && g < 5) load_row(spn, g + 1, dr[(g + 1) & 1]);
            if constexpr (s == 3) issue_piece(WInt<g>{});
#ifdef W4W_STAMP
            if constexpr (s == 7) asm volatile("s_memtime %0" : "=s"(ts[2 + g]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        w4w_for<0, 72>(slot);
        issue_tail();
        rslot = nslot;
#ifdef W4W_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (blockIdx.x == 0 && kcnt < 64 && lane < 11) {
            unsigned long long tv = 0;
#pragma unroll
            for (int i = 0; i < 11; ++i) tv = lane == i ? ts[i] : tv;
            w4w_stamps[(wave * 64 + kcnt) * 12 + lane] = tv;
        }
        ++kcnt;
#endif
    };

    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        for (int cs = 0; cs < nk; cs += 2) {
            kstep(WInt<0>{}, v, w);          // (nks is even: the launcher)
            kstep(WInt<1>{}, w, v);
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // MFMA result -> reader: hipcc does not see the MFMAs inside the statements above
        // ---- epilogue of task ct (conv_wino4.h's, for both row halves) ---------------------------------------------------------
        int crt, tile;
        task_of(ct, crt, tile);
        const int ttx = tile % p.ntx, tty = (tile / p.ntx) % p.nty, b = tile / (p.ntx * p.nty);
        const int y = tty * TS + 4 * tyl, x = ttx * TS + 4 * tx;
        const int rW = p.W >> p.res_up, rHW = rW * (p.H >> p.res_up);
        auto epi = [&](auto HIt) {
            constexpr int hi = decltype(HIt)::value, h = hi >> 2, i = hi & 3;
            auto M = [&](auto Xt) -> float {
                constexpr int ai = 36 * h + decltype(Xt)::value;
                if constexpr (ai < 64) return accA[ai][i];
                else return accV[ai - 64][i];
            };
            const int row = crt * 32 + h * 16 + 4 * kk + i, rc = row < p.Cout ? row : p.Cout - 1;
            const float bsv = p.bias ? p.bias[rc] : 0.f;
            f32x4 rr[4];                                 // residual of the (row, block): loaded first, consumed after the transforms
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
            auto col = [&](auto Jt) {
                constexpr int j = decltype(Jt)::value;
                wino4_out1d(M(WInt<j>{}), M(WInt<6 + j>{}), M(WInt<12 + j>{}), M(WInt<18 + j>{}), M(WInt<24 + j>{}), M(WInt<30 + j>{}), t[0][j], t[1][j],
                            t[2][j], t[3][j]);
            };
            w4w_for<0, 6>(col);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o0, o1, o2, o3;
                wino4_out1d(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], o0, o1, o2, o3);
                const f32x4 o = {o0 + bsv + rr[r].x, o1 + bsv + rr[r].y, o2 + bsv + rr[r].z, o3 + bsv + rr[r].w};
                if (row < p.Cout) *reinterpret_cast<f32x4*>(p.out + ((long long)b * p.Cout + row) * HW + (y + r) * p.W + x) = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        w4w_for<0, 8>(epi);
#pragma unroll
        for (int x2 = 0; x2 < 64; ++x2) accA[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x2 = 0; x2 < 8; ++x2) accV[x2] = (f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the epilogue's loads / stores share the counter with the ring: drain once per task
    }
}

}  // namespace chk
