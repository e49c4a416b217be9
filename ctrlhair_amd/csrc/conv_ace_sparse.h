// conv_ace_sparse.h -- the SPADE gamma/beta conv + fused ACE epilogue of conv_mfma.h (exact-f32 MFMA,
// v_mfma_f32_32x32x2_f32) over the BOUNDARY pixels only (ace_sparse.h): the pixels of a 32 x TH tile whose 5x5 label
// neighbourhood is not uniform are compacted into 32-pixel sub-tiles; interior pixels never reach a matrix core.
//
// Same GEMM view, weight packing (pack_A, CK = 16) and LDS staging of the un-expanded input patch as conv_mfma_kernel; what
// changes is the N side: a lane's B operand is read at the patch offset of ITS compacted pixel (ds_read_b32 with per-lane
// addresses was already the access pattern), and a wave owns 64 rows x NSUB <= 4 sub-tiles.  A block = 4 wave tasks of one
// spatial tile (they share the staged patch): with mtiles % 4 == 0 the four waves run the same sub-tile group against four
// different 64-row tiles.  The grid is an upper bound (dense case); blocks beyond the device-side task count exit.
#pragma once
#include "ace_sparse.h"
#include "conv_mfma.h"

namespace chk {

template <int TH>
struct SpCfg {
    static constexpr int TW = 32, CK = 16, PW = TW + 2, PH = TH + 2, PLANE = PH * PW;
    static constexpr int SE = CK * PLANE, NLOAD = (SE + 255) / 256, KSTEPS = 9 * CK / 2, NG = KSTEPS / 4;
    static constexpr int LDS_BYTES = 2 * SE * 4;
};

// the work of one wave: NSUB = 0 -> staging and barriers only (idle wave of a partially filled block)
template <int TH, int NSUB>
__device__ __forceinline__ void ace_sparse_body(const ConvParams& p, float* smem, int b0, int y0, int x0, int mtile64, int s0,
                                                int cnt, const uint16_t* __restrict__ lst) {
    using Cfg = SpCfg<TH>;
    constexpr int PW = Cfg::PW, PLANE = Cfg::PLANE, NG = Cfg::NG, NLOAD = Cfg::NLOAD, SE = Cfg::SE, CK = Cfg::CK;
    constexpr int NA = NSUB > 0 ? NSUB : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int HW = p.H * p.W;

    // compacted pixels of this wave's sub-tiles: in-tile offset ty*32+tx; slots beyond cnt repeat the last boundary pixel
    int pix_[NA], loff[NA];
    bool pok_[NA];
#pragma unroll
    for (int n = 0; n < NSUB; ++n) {
        const int slot = (s0 + n) * 32 + (lane & 31);
        pok_[n] = slot < cnt;
        pix_[n] = lst[pok_[n] ? slot : cnt - 1];
        loff[n] = (lane >> 5) * PLANE + (pix_[n] >> 5) * PW + (pix_[n] & 31);
    }

    f32x16 acc[2][NA];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NA; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    auto stage = [&](int chunk, int buf) {
        float stg[NLOAD];
        const float* src = p.in + ((long long)b0 * p.Cin + (long long)chunk * CK) * HW;
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            int e = tid + i * 256;
            asm volatile("" : "+v"(e));     // keep the decode inside the chunk loop (no hoisted address regs)
            float v = 0.f;
            if (e < SE) {
                const int c = e / PLANE, rem = e % PLANE;
                const int py = rem / PW, px = rem % PW;
                const int y = y0 - 1 + py, x = x0 - 1 + px;
                const bool ok = chunk * CK + c < p.Cin && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                if (ok) v = src[c * HW + y * p.W + x];
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

    const float4* Ap = reinterpret_cast<const float4*>(p.wpk) + ((long long)mtile64 * p.nchunks) * (NG * 2 * 64) + lane;

    // sean.dbg bit 256 (profiling only): lane 0 of every wave stamps s_memtime at block start / after the prologue barrier /
    // after the k-loop / after the epilogue, plus its NSUB (tools/sparse_timeline.py)
    const bool stamp = (p.dbg & 256) && p.partial && lane == 0;
    long long* stamps = reinterpret_cast<long long*>(p.partial) + ((long long)blockIdx.x * 4 + (tid >> 6)) * 5;
    if (stamp) { stamps[0] = __builtin_amdgcn_s_memtime(); stamps[4] = NSUB; }
    stage(0, 0);
    __syncthreads();
    if (stamp) stamps[1] = __builtin_amdgcn_s_memtime();
    for (int ch = 0; ch < p.nchunks; ++ch) {
        if (ch + 1 < p.nchunks) stage(ch + 1, (ch + 1) & 1);
        if constexpr (NSUB > 0) {
            const float* sb = smem + (ch & 1) * SE;
            const float4* Ac = Ap + (long long)ch * (NG * 2 * 64);
            float4 a0 = Ac[0], a1 = Ac[64];
            float bv[NA], bn[NA];
#pragma unroll
            for (int n = 0; n < NSUB; ++n) bv[n] = sb[loff[n]];
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
                    const int s1 = g * 4 + q + 1;
                    if (s1 < Cfg::KSTEPS) {
                        const int t = s1 / HALF, cp = s1 % HALF;
                        const int koff = 2 * cp * PLANE + (t / 3) * PW + (t % 3);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int n = 0; n < NSUB; ++n) bn[n] = sb[koff + loff[n]];
                    }
#pragma unroll
                    for (int n = 0; n < NSUB; ++n) {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[q], bv[n], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[q], bv[n], acc[1][n], 0, 0, 0);
                    }
#pragma unroll
                    for (int n = 0; n < NSUB; ++n) bv[n] = bn[n];
                }
                a0 = a0n;
                a1 = a1n;
            }
        }
        __syncthreads();
    }

    if (stamp) stamps[2] = __builtin_amdgcn_s_memtime();
    // ---- ACE epilogue (conv_mfma.h ace_epilogue_f32) on the compacted pixels
    if constexpr (NSUB > 0) {
        int py[NA], px[NA];
#pragma unroll
        for (int n = 0; n < NSUB; ++n) {
            py[n] = y0 + (pix_[n] >> 5);
            px[n] = x0 + (pix_[n] & 31);
        }
        ace_epilogue_f32<NA>(p, acc, mtile64, lane >> 5, b0, py, px, pok_);
    }
    if (stamp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (profiling only) include the store drain
        stamps[3] = __builtin_amdgcn_s_memtime();
    }
}

template <int TH>
__global__ __launch_bounds__(256, 2) void conv_ace_sparse_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int total = p.sp_total[0];
    if ((int)blockIdx.x >= total) return;
    const int L = xcd_remap(blockIdx.x, total);
    const unsigned wk = p.sp_work[L];
    const int tile = wk & 0xFFFFF, bt = wk >> 20;
    const int tpi = p.tiles_x * p.tiles_y;
    const int b0 = tile / tpi, tr = tile - b0 * tpi;
    const int y0 = (tr / p.tiles_x) * TH, x0 = (tr % p.tiles_x) * 32;
    const int cnt = p.sp_cnt[tile], NS = (cnt + 31) >> 5;
    int ng, per;
    sparse_groups(NS, p.mtiles, ng, per);
    const int t = bt * 4 + (threadIdx.x >> 6);
    const int g = t / p.mtiles, mtile64 = t - g * p.mtiles;
    int nsub = 0, s0 = 0;
    if (g < ng) {
        s0 = g * per;
        nsub = NS - s0 < per ? NS - s0 : per;
    }
    const uint16_t* lst = p.sp_list + (long long)tile * (32 * TH);
    switch (nsub) {            // wave-uniform; every variant executes the same barriers
        case 0: ace_sparse_body<TH, 0>(p, smem, b0, y0, x0, mtile64, s0, cnt, lst); break;
        case 1: ace_sparse_body<TH, 1>(p, smem, b0, y0, x0, mtile64, s0, cnt, lst); break;
        case 2: ace_sparse_body<TH, 2>(p, smem, b0, y0, x0, mtile64, s0, cnt, lst); break;
        case 3: ace_sparse_body<TH, 3>(p, smem, b0, y0, x0, mtile64, s0, cnt, lst); break;
        default: ace_sparse_body<TH, 4>(p, smem, b0, y0, x0, mtile64, s0, cnt, lst); break;
    }
}

// p.sp_* must be set (level of p.H x p.W classified with tile height TH; work list built for rows / 64 row tiles)
hipError_t conv_ace_sparse(const ConvParams& p, int TH, hipStream_t s);

}  // namespace chk
