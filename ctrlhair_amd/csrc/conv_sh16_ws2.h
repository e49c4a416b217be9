// conv_sh16_ws2.h -- EXPERIMENT (selectable with option sean.dbg bit 2048, covered by the parity tests, NOT the default):
// the SPADE gamma/beta conv (3x3, Cin = 128) with the fused ACE epilogue, wave-specialised and persistent like
// conv_sh16_ws_kernel, but with the EPILOGUE OF TILE k SOFTWARE-PIPELINED INTO THE K LOOP OF TILE k+1.
//
// Motivation (profiles/r02_ws_timeline.md): in conv_sh16_ws_kernel a tile's k-loop runs the matrix cores at 91 % of their
// issue rate, but the epilogue that follows leaves them idle for 22 % of every tile.  Two accumulator sets per consumer
// wave would hide it -- and with 256 registers per wave that means HALF the pixels per wave (64 rows x 64 px: 4
// accumulators = 64 registers): while set `cur` accumulates tile k+1, the finished set `prv` of tile k is drained in 8
// steps of (pixel sub-tile, 4-channel run), one step per 16-channel chunk of the k-loop:
//      chunk top  : x load of the step and the style-LUT gathers of filter tap 0
//      tap t < 8  : the gathers of filter tap t are accumulated, those of tap t+1 issued
//      tap 8      : last accumulation, then modulation / re-split / store under the tap's MFMAs (whose operand-prefetch
//                   registers are free: there is no tap 9)
// Tiles are 32 x 8 pixels (patch 34 x 10), LDS: 2 stages x (21.8 KB patch + 36 KB A fragments) + two copies (tile parity)
// of the small epilogue operands.  Loaders: as in conv_sh16_ws_kernel.
//
// Result (same file): correct on every parity test, and NOT faster.  Halving the pixels per tile doubles the A-fragment
// bytes per MFMA (58 KB per 3456 matrix-core cycles = 16.8 B/cycle/CU instead of 75 KB per 6912 = 10.9); the loaders
// deliver ~12 B/cycle/CU, so the k-loop becomes delivery-bound: 37.9 k cycles per 256-pixel tile (27.6 k ideal; 29.5 k
// with loader traffic and epilogue both switched off, 33.3 k with the epilogue woven in but no loader traffic, 38.5 k
// with loader traffic and no epilogue) -- 75.8 k per 512 pixels against 77.7 k for the kernel it was meant to beat.
// The design needs operand delivery that scales with tile count (A fragments shared across CUs / multicast), which this
// hardware generation does not offer to a HIP kernel; kept for the record and for the next attempt.
#pragma once
#include "conv_sh16.h"

namespace chk {

template <int TERMS, bool STYLED>
__global__ __launch_bounds__(512, 2) void conv_sh16_ws2_kernel(const ConvParams p) {
    constexpr int TW = 32, TH = 8, PW = TW + 2, PH = TH + 2, PLANE = PW * PH, UNITS = 4 * PLANE;     // 1360 units
    constexpr int NLD = (UNITS + 255) / 256;                                                          // 6
    constexpr int AUNITS = 9 * 4 * 64, STAGE = UNITS + AUNITS, NDA = AUNITS / 256;
    constexpr int NPAR = 5, LW = TW + 2, LH = TH + 2;
    constexpr int SM_NZ = 8 * NPAR, SM_LAB = SM_NZ + (TW * TH) / 4, SMALL = SM_LAB + (LH * LW + 15) / 16;
    constexpr int SM0 = 2 * STAGE;
    constexpr int NCH = 8;                                    // Cin == 128: 8 chunks of 16 channels = 8 epilogue steps
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];

    sh16_mode_on();                                           // saturating f32 -> f16 conversions in the epilogue
    if (p.pass == 1 && sh16_dyn_extra(*p.out_amax) == 1.f) return;     // second pass: nothing to repair (the normal case)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool loader = wave >= 4;
    const int wn = wave & 3, ltid = tid & 255;
    const int HW = p.H * p.W;
    const int G = p.Cin >> 3;
    const int ntiles = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    const int first = xcd_remap(blockIdx.x, gridDim.x);
    const int my_tiles = first < ntiles ? (ntiles - 1 - first) / (int)gridDim.x + 1 : 0;
    const int Q = my_tiles * NCH;
    const uint4* gin = reinterpret_cast<const uint4*>(p.in);

    auto tile_coords = [&](int k, int& mtile64, int& x0, int& y0, int& b0) {
        const int L = first + k * (int)gridDim.x;
        mtile64 = L % p.mtiles;
        int nt = L / p.mtiles;
        const int txi = nt % p.tiles_x; nt /= p.tiles_x;
        const int tyi = nt % p.tiles_y; nt /= p.tiles_y;
        x0 = txi * TW; y0 = tyi * TH; b0 = nt;
    };

    if (loader) {
        // ---------------------------------------------------------------- loaders (see conv_sh16_ws_kernel)
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
                    const int py = rem / PW, px = rem % PW;
                    const int y = y0 + py - 1, x = x0 + px - 1;
                    if (b0 < p.B && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
                        soff[i] = ((b0 * G + (gh >> 1)) * 2 + (gh & 1)) * HW + y * p.W + x;
                    if (TERMS != 3 && (gh & 1)) soff[i] = -1;           // single-term path never reads the lo planes
                }
            }
            cur_tile = k;
        };
        uint4 stgA[NLD], stgB[NLD];
        auto load_chunk = [&](int q, uint4 (&stg)[NLD]) {
            const int k = q / NCH, ch = q % NCH;
            if (k != cur_tile) set_tile(k);
            const uint4* src = gin + (long long)ch * 4 * HW;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {                          // exactly NLD loads, always (vmcnt bookkeeping below)
                const uint4 v = src[soff[i] >= 0 ? soff[i] : 0];
                stg[i] = soff[i] >= 0 ? v : make_uint4(0, 0, 0, 0);
            }
        };
        const uint4* gA = reinterpret_cast<const uint4*>(p.wpk);
        auto dma_A = [&](int q) {
            int mt, x0, y0, b0;
            tile_coords(q / NCH, mt, x0, y0, b0);
            const uint4* src = gA + ((long long)mt * NCH + q % NCH) * AUNITS + ltid;
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
        // small epilogue operands of tile k -> copy (k & 1): read by the consumers during tile k+1
        float4 parr = make_float4(0.f, 0.f, 0.f, 0.f);
        float nzr = 0.f;
        uint8_t labr[2] = {255, 255};
        auto epi_load = [&](int k) {
            int mt, x0, y0, b0;
            tile_coords(k, mt, x0, y0, b0);
            const int C = p.C;
            if (ltid < 8 * NPAR) {                                        // par[run][which]
                const int run = ltid / NPAR, which = ltid % NPAR;
                const int c0 = (mt * 8 + run) * 4;
                const float* src = which == 0 ? p.bias_g : (which == 1 ? p.bias_b : (which == 2 ? p.bn_a : (which == 3 ? p.bn_d : p.nv)));
                parr = *reinterpret_cast<const float4*>(src + (c0 < C ? c0 : 0));
                const float osc = p.out_scale != 0.f ? p.out_scale : 1.f;      // folded output scale (ace_quad)
                if (which == 0) parr = make_float4((parr.x + 1.f) * osc, (parr.y + 1.f) * osc, (parr.z + 1.f) * osc, (parr.w + 1.f) * osc);
                if (which == 1) parr = make_float4(parr.x * osc, parr.y * osc, parr.z * osc, parr.w * osc);
            }
            {
                const int tx = ltid % TW, ty = ltid / TW;
                const int y = y0 + ty, x = x0 + tx;
                const bool ok = y < p.H && x < p.W;
                nzr = p.noise[ok ? (long long)b0 * p.noise_bstride + (long long)x * p.H + y : 0];
            }
            if (STYLED) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = ltid + i * 256;
                    labr[i] = 255;
                    if (e < LH * LW) {
                        const int ly = e / LW, lx = e % LW;
                        const int y = y0 - 1 + ly, x = x0 - 1 + lx;
                        const bool in = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                        const uint8_t v = p.lab[in ? (long long)b0 * HW + y * p.W + x : 0];
                        labr[i] = in ? v : (uint8_t)255;
                    }
                }
            }
        };
        auto epi_store = [&](int k) {
            uint4* sm = smem_u + SM0 + (k & 1) * SMALL;
            if (ltid < 8 * NPAR) reinterpret_cast<float4*>(sm)[ltid] = parr;
            reinterpret_cast<float*>(sm + SM_NZ)[ltid] = nzr;
            if (STYLED) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int e = ltid + i * 256;
                    if (e < LH * LW) reinterpret_cast<uint8_t*>(sm + SM_LAB)[e] = labr[i];
                }
            }
        };
        if (Q > 0) {
            dma_A(0);
            load_chunk(0, stgA);
            if (Q > 1) load_chunk(1, stgB);
            store_chunk(0, stgA);
            if (Q > 2) load_chunk(2, stgA);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                      // stage 0 ready
        auto iter = [&](int q, uint4 (&stg)[NLD]) {
            const int k = q / NCH, ch = q % NCH;
            if (q + 1 < Q && !(p.dbg & 1)) {
                dma_A(q + 1);
                store_chunk((q + 1) & 1, stg);
            }
            if (ch == NCH - 1) epi_store(k);                  // consumers read these after this iteration's barrier
            if (q + 3 < Q && !(p.dbg & 1)) {
                load_chunk(q + 3, stg);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (ch == NCH - 2) epi_load(k);                   // younger than everything waited on above
            __syncthreads();
        };
        for (int q = 0; q < Q; q += 2) {
            iter(q, stgB);
            if (q + 1 < Q) iter(q + 1, stgA);
        }
        return;
    }

    // -------------------------------------------------------------------- consumers
    const int hi = lane >> 5, col = lane & 31;
    int ub[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) ub[n] = hi * 2 * PLANE + (wn * 2 + n) * PW + col;
    f32x16 cur[2][2], prv[2][2];                              // [M-subtile: gamma | beta rows][pixel sub-tile]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) { cur[m][n][r] = 0.f; prv[m][n][r] = 0.f; }

    const int C = p.C, Go = (C + 7) >> 3;
    const int xW = p.W >> p.x_up, xHW = xW * (p.H >> p.x_up);
    const unsigned lrs = (unsigned)p.lut_rs * 4u, lns = (unsigned)p.lut_ns * 4u;          // LUT strides in bytes
    const float slope = act_slope(p.act);
    const float sc0 = (p.in_scale_inv != 0.f ? p.in_scale_inv : 1.f) * (p.out_scale != 0.f ? p.out_scale : 1.f);
    const float extra = p.pass == 1 ? sh16_dyn_extra(*p.out_amax) : 1.f;
    float amax = 0.f;

    // ---- state of the tile being drained (tile k-1 while tile k accumulates)
    bool pvalid[2] = {false, false};                          // this lane's pixel of sub-tile n exists (and there is a tile)
    unsigned poo[2] = {0, 0}, pxo[2] = {0, 0};                // byte offsets of the pixel in the output / x tensors
    float pnz[2] = {0.f, 0.f};
    int pmt = 0;
    const char* xbase = reinterpret_cast<const char*>(p.x);
    char* obase = reinterpret_cast<char*>(p.out);
    const char* lbase = reinterpret_cast<const char*>(p.lut);
    const uint4* small = smem_u + SM0;
    float st = sc0;
    // ---- registers of the epilogue step in flight (a step lives entirely inside one chunk of the k-loop)
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;     // style sums of the step
    float4 slot_g = sg, slot_b = sg;                          // the one gather pair in flight
    float4 xq = sg;                                           // x of the step (loaded at tap 0, used at tap 8)
    unsigned long long jv = 0;                                // 9 labels x 5 bits of the pixel's 3x3 neighbourhood (19 = zero column)
    uint4 wq = make_uint4(0, 0, 0, 0);                        // the step's split result between its two halves

    auto set_prev = [&](int k) {          // geometry of tile k becomes the drained tile's
        int x0, y0, b0;
        tile_coords(k, pmt, x0, y0, b0);
        xbase = reinterpret_cast<const char*>(p.x) + (long long)b0 * (C >> 2) * xHW * 16;
        obase = reinterpret_cast<char*>(p.out) + (long long)b0 * Go * 2 * HW * 16;
        lbase = reinterpret_cast<const char*>(p.lut) + (long long)b0 * p.lut_bs * lns;
        small = smem_u + SM0 + (k & 1) * SMALL;
        st = (p.wscale ? p.wscale[pmt * 64] : 1.f) * sc0;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int ty = wn * 2 + n, y = y0 + ty, x = x0 + col;
            pvalid[n] = y < p.H && x < p.W;
            const int yc = pvalid[n] ? y : 0, xc = pvalid[n] ? x : 0;
            poo[n] = ((unsigned)hi * (unsigned)HW + (unsigned)yc * p.W + (unsigned)xc) * 16u;
            pxo[n] = ((unsigned)(yc >> p.x_up) * xW + (unsigned)(xc >> p.x_up)) * 16u;
        }
    };
    // step s = (pixel sub-tile n = s / 4, channel run rq = s % 4)
    auto step_cc = [&](int s, bool& cok) {
        const int c0 = (pmt * 4 + (s & 3)) * 8 + 4 * hi;
        cok = c0 < C;
        return (unsigned)(cok ? c0 : 0);
    };
    auto gather_issue = [&](int s, int t) {
        if (!STYLED) return;
        bool cok;
        const unsigned cc = step_cc(s, cok);
        const unsigned j = t < 6 ? ((unsigned)jv >> (5 * t)) & 31u : ((unsigned)(jv >> 30) >> (5 * (t - 6))) & 31u;
        const unsigned o1 = cc * lrs + j * lns + (unsigned)(t * 2 * C) * lrs;
        slot_g = *reinterpret_cast<const float4*>(lbase + o1);
        slot_b = *reinterpret_cast<const float4*>(lbase + (o1 + (unsigned)C * lrs));
    };
    auto gather_acc = [&]() {
        if (!STYLED) return;
        sg.x += slot_g.x; sg.y += slot_g.y; sg.z += slot_g.z; sg.w += slot_g.w;
        sb.x += slot_b.x; sb.y += slot_b.y; sb.z += slot_b.z; sb.w += slot_b.w;
    };
    auto step_begin = [&](int s) {        // top of the step's chunk: x load, labels / noise of the pixel, first gather
        const int n = s >> 2;
        bool cok;
        const unsigned cc = step_cc(s, cok);
        xq = *reinterpret_cast<const float4*>(xbase + (pxo[n] + (cc >> 2) * (unsigned)xHW * 16u));
        if ((s & 3) == 0) {
            pnz[n] = reinterpret_cast<const float*>(small + SM_NZ)[wn * 64 + n * 32 + col];
            if (STYLED) {
                const uint8_t* lp = reinterpret_cast<const uint8_t*>(small + SM_LAB) + (wn * 2 + n) * LW + col;
                unsigned long long lv = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const unsigned j = lp[(t / 3) * LW + (t % 3)];
                    lv |= (unsigned long long)(j < 19u ? j : 19u) << (5 * t);
                }
                jv = lv;
            }
        }
        sg = make_float4(0.f, 0.f, 0.f, 0.f);
        sb = sg;
        gather_issue(s, 0);
    };
    // modulation of step s in two halves (2 channels each): gamma / beta from the drained accumulators, ACE, re-split
    auto finalize_half = [&](auto sc, auto hc) {
        constexpr int s = decltype(sc)::value, h = decltype(hc)::value, n = s >> 2, rq = s & 3;
        const int run = rq * 2 + hi;
        const float* par = reinterpret_cast<const float*>(small) + run * NPAR * 4 + 2 * h;      // [which][4 channels]
        const f32x2 vpg = *reinterpret_cast<const f32x2*>(par + 0), vpb = *reinterpret_cast<const f32x2*>(par + 4),
                    va = *reinterpret_cast<const f32x2*>(par + 8), vd = *reinterpret_cast<const f32x2*>(par + 12),
                    vn = *reinterpret_cast<const f32x2*>(par + 16);
        const f32x2 ag = {prv[0][n][rq * 4 + 2 * h], prv[0][n][rq * 4 + 2 * h + 1]};
        const f32x2 ab = {prv[1][n][rq * 4 + 2 * h], prv[1][n][rq * 4 + 2 * h + 1]};
        const f32x2 vsg = h ? f32x2{sg.z, sg.w} : f32x2{sg.x, sg.y}, vsb = h ? f32x2{sb.z, sb.w} : f32x2{sb.x, sb.y};
        const f32x2 vx = h ? f32x2{xq.z, xq.w} : f32x2{xq.x, xq.y};
        const f32x2 gam1 = ag * st + (vpg + vsg);
        const f32x2 bet = ab * st + (vpb + vsb);
        const f32x2 nrm = va * vx + (vn * pnz[n] + vd);
        f32x2 o = nrm * gam1 + bet;
        const f32x2 os = o * slope;
        o.x = fmaxf(o.x, os.x);
        o.y = fmaxf(o.y, os.y);
        o = o * extra;                                        // 1 except in a repair pass (sh16.h)
        bool cok;
        (void)step_cc(s, cok);
        const float am = fmaxf(fmaxf(amax, fabsf(o.x)), fabsf(o.y));
        if (pvalid[n] && cok) amax = am;                      // (no tile yet / padding channels: not recorded)
        if constexpr (TERMS == 2) {
            const bf16x2 hb = __builtin_convertvector(o, bf16x2);
            if (h == 0) { wq.x = __builtin_bit_cast(unsigned, hb); wq.z = 0u; }
            else        { wq.y = __builtin_bit_cast(unsigned, hb); wq.w = 0u; }
        } else {
            const f16x2 hh = __builtin_convertvector(o, f16x2);
            const f32x2 lo = o - __builtin_convertvector(hh, f32x2);
            const f16x2 ll = __builtin_convertvector(lo, f16x2);
            if (h == 0) { wq.x = __builtin_bit_cast(unsigned, hh); wq.z = __builtin_bit_cast(unsigned, ll); }
            else        { wq.y = __builtin_bit_cast(unsigned, hh); wq.w = __builtin_bit_cast(unsigned, ll); }
        }
    };
    auto finalize_store = [&](auto sc) {
        constexpr int s = decltype(sc)::value, n = s >> 2, rq = s & 3;
        bool cok;
        (void)step_cc(s, cok);
        const int g = pmt * 4 + rq;
        uint4 w = wq;
        if (!cok) w = make_uint4(0, 0, 0, 0);                 // padding channels of the last group hold zeros
        w = sh16_pair_swap(w);
        if (pvalid[n] && g < Go) *reinterpret_cast<uint4*>(obase + (poo[n] + (unsigned)g * 2u * (unsigned)HW * 16u)) = w;
    };

    __syncthreads();                                          // stage 0 ready
    int q = 0;
    const bool stamp = (p.dbg & 256) && wn == 0 && lane == 0 && p.partial;      // profiling only (tools/ws_timeline.py)
    long long* stamps = reinterpret_cast<long long*>(p.partial) + (long long)blockIdx.x * 64 * 3;
    for (int k = 0; k < my_tiles; ++k) {
        if (stamp && k < 64) stamps[k * 3] = __builtin_amdgcn_s_memtime();
        // ---- k-loop of tile k with the drain of tile k-1 woven in (all step / tap indices are compile-time).
        // Step `ch` lives in chunk `ch`: x load + first gather at the top; LUT tap u is gathered during MFMA tap u-1 and
        // accumulated during MFMA tap u; the modulation runs under the MFMAs of tap 8, whose operand-prefetch registers
        // are free (there is no tap 9 to fetch).
        auto chunk = [&](auto chc) {
            constexpr int ch = decltype(chc)::value;
            const uint4* sbp = smem_u + (q & 1) * STAGE;
            const uint4* sa = sbp + UNITS + lane;
            uint4 a_cur[4], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a_cur[i] = sa[i * 64];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bh[n] = sbp[ub[n]];
                bl[n] = sbp[ub[n] + PLANE];
            }
            const bool epi = !(p.dbg & 4);
            if (epi) step_begin(ch);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                uint4 a_nxt[4], bhn[2], bln[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) a_nxt[i] = a_cur[i];
#pragma unroll
                for (int n = 0; n < 2; ++n) { bhn[n] = bh[n]; bln[n] = bl[n]; }
                const int koff = ((t + 1) / 3) * PW + ((t + 1) % 3);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    // group i: operand fetches of tap t+1 / epilogue micro-ops, then 2 MFMAs of tap t
                    if (t + 1 < 9) {
                        if (i == 0) { a_nxt[0] = sa[((t + 1) * 4 + 0) * 64]; a_nxt[1] = sa[((t + 1) * 4 + 1) * 64]; }
                        if (i == 1) { a_nxt[2] = sa[((t + 1) * 4 + 2) * 64]; a_nxt[3] = sa[((t + 1) * 4 + 3) * 64]; }
                        if (i == 2) { bhn[0] = sbp[ub[0] + koff]; bln[0] = sbp[ub[0] + koff + PLANE]; }
                        if (i == 3) { bhn[1] = sbp[ub[1] + koff]; bln[1] = sbp[ub[1] + koff + PLANE]; }
                        if (i == 4 && epi) gather_acc();                   // LUT tap t (issued during tap t-1 / at the top)
                        if (i == 5 && epi) gather_issue(ch, t + 1);
                    } else if (epi) {
                        if (i == 0) gather_acc();                          // LUT tap 8
                        if (i == 1) finalize_half(chc, std::integral_constant<int, 0>{});
                        if (i == 2) finalize_half(chc, std::integral_constant<int, 1>{});
                        if (i == 3) finalize_store(chc);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * i + jj, term = j >> 2, m = (j & 3) >> 1, n = j & 1;
                        if (TERMS != 3 && term != 2) continue;
                        cur[m][n] = mfma16<TERMS>(a_cur[m * 2 + (term == 0 ? 1 : 0)], term == 1 ? bl[n] : bh[n], cur[m][n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
                for (int n = 0; n < 2; ++n) { bh[n] = bhn[n]; bl[n] = bln[n]; }
            }
            __syncthreads();
            ++q;
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        chunk(std::integral_constant<int, 4>{});
        chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{});
        chunk(std::integral_constant<int, 7>{});
        if (stamp && k < 64) stamps[k * 3 + 1] = stamps[k * 3 + 2] = __builtin_amdgcn_s_memtime();
        // tile k is complete: it becomes the drained tile
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                prv[m][n] = cur[m][n];
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[m][n][r] = 0.f;
            }
        set_prev(k);
    }
    // ---- drain of the last tile (no k-loop to hide under)
    if (my_tiles > 0) {
        auto tail = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            step_begin(s);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                gather_acc();
                if (t + 1 < 9) gather_issue(s, t + 1);
            }
            finalize_half(sc, std::integral_constant<int, 0>{});
            finalize_half(sc, std::integral_constant<int, 1>{});
            finalize_store(sc);
        };
        tail(std::integral_constant<int, 0>{});
        tail(std::integral_constant<int, 1>{});
        tail(std::integral_constant<int, 2>{});
        tail(std::integral_constant<int, 3>{});
        tail(std::integral_constant<int, 4>{});
        tail(std::integral_constant<int, 5>{});
        tail(std::integral_constant<int, 6>{});
        tail(std::integral_constant<int, 7>{});
    }
    if (p.pass != 1 && p.out_amax) {                           // one atomic per consumer wave per launch
        amax = sh16_wave_max(amax);
        if (lane == 0) sh16_slot_max(p.out_amax, amax);
    }
}

template <int TERMS>
hipError_t launch_sh16_ws2(ConvParams p, int rows, hipStream_t stream) {
    if (p.Cin != 128 || p.in_mode != IN_DIRECT) return hipErrorInvalidValue;
    constexpr int UNITS = 4 * 34 * 10, STAGE = UNITS + 9 * 4 * 64, SMALL = 40 + 64 + (10 * 34 + 15) / 16;
    constexpr int LDS = (2 * STAGE + 2 * SMALL) * 16;
    static bool attr_set[64] = {};                           // per device (a process may own handles on several GPUs)
    static int ncus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sh16_ws2_kernel<TERMS, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sh16_ws2_kernel<TERMS, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        ncus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        attr_set[dev] = true;
    }
    const int ncu = ncus[dev];
    p.nchunks = 8;
    p.mtiles = (rows + 63) / 64;
    p.tiles_x = (p.W + 31) / 32;
    p.tiles_y = (p.H + 7) / 8;
    p.tiles_b = p.B;
    p.splitk = 1;
    const int ntiles = p.mtiles * p.tiles_x * p.tiles_y * p.tiles_b;
    const int grid = ntiles < ncu ? ntiles : ncu;
    if (p.lut) hipLaunchKernelGGL((conv_sh16_ws2_kernel<TERMS, true>), dim3(grid), dim3(512), LDS, stream, p);
    else hipLaunchKernelGGL((conv_sh16_ws2_kernel<TERMS, false>), dim3(grid), dim3(512), LDS, stream, p);
    return hipGetLastError();
}

}  // namespace chk
