// conv_pw.h -- 1x1 ("pointwise") convolution on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32) for the learned ResBlock
// shortcut conv_s (architecture.py:75-79: x_s = conv_s(ace_s(x)), no bias) of the exact-f32 Winograd path, whose output the
// Winograd conv_1 launch adds as its residual (conv_wino.h).
//
// A 1x1 conv is a plain GEMM  out[row][pixel] = sum_ci W[row][ci] in[ci][pixel]  with 9x less reuse of every staged input byte
// than a 3x3 conv, so the block shares its input tile among RT row tiles: 8 waves = RT row tiles of 32 rows x (8 / RT) groups of
// 64 pixels; one ring stage = 16 input channels = 8 k-steps = 16 MFMAs (1024 matrix-pipe cycles) per wave against 16 KB
// (RT = 4: 128 pixels, 8 KB of input + 8 KB of A) or 20 KB (RT = 2: 256 pixels) of LDS-DMA.  Same machinery as conv_wino.h:
// persistent 512-thread blocks, every operand by LDS-DMA (buffer_load ... lds, 16 B per lane -- the pixel rows of a 1x1 conv are
// aligned, there is no halo), a 6-stage ring running as one flat sequence across the block's tasks, one barrier per stage.
// No input transform and no per-tap addressing: the non-MFMA instruction stream is ~40 instructions per 16 MFMAs.
#pragma once
#include "conv_wino.h"

namespace chk {

// GROUPED launch (conv_pw_grouped): several GEMMs with the same Cin and Cout but their own operands and pixel counts run as ONE
// persistent launch -- the exact-f32 style-LUT builds of all styled ACEs of the generator (sean_model.cpp: operands swapped, the
// "pixels" are the 18 C rows of conv_gamma / conv_beta, the GEMM rows the (sample, label) columns).  Group g owns the tasks
// [start, next group's start) x row groups: local task -> (row group fastest, pixel tile).
struct PwGroup {
    const float* in;        // [Cin][HW]
    const float* wpk;       // pack_pw_A image of the group's Cout rows
    float* out;             // [Cout][HW]
    int HW, start;          // HW % 128 == 0; pixel tiles (of 128) before this group: its first task is start * (row groups of the launch)
};
struct PwParams {
    const float* in;        // [B][Cin][HW]
    const float* wpk;       // pack_pw_A image
    float* out;             // [B][Cout][HW]
    int B, Cin, Cout, HW;   // Cin % 16 == 0, HW % (512 / RT) == 0
    const PwGroup* groups;  // device array of ngroups entries, or null (plain launch)
    int ngroups;
    long long in_bs;        // plain launch: floats between the samples of `in`; 0 = Cin * HW (a launch over the first Cin planes of wider samples: sean_model.cpp, Zencoder ConvTranspose)
    // set by the launcher
    int nrg, npt, ntasks, nst;
};

// image of (row tile rt, stage st): [lane][8 k-steps] floats: W[row = 32 rt + (lane & 31)][ci = 16 st + 2 s + (lane >> 5)]
template <class F>
std::vector<float> pack_pw_A(int rows, int Cin, F get) {
    const int nrt = (rows + 31) / 32, nst = Cin / 16;
    std::vector<float> dst((size_t)nrt * nst * 512, 0.f);
    for (int rt = 0; rt < nrt; ++rt)
        for (int st = 0; st < nst; ++st)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 8; ++s) {
                    const int row = rt * 32 + (lane & 31), ci = 16 * st + 2 * s + (lane >> 5);
                    if (row < rows) dst[((size_t)rt * nst + st) * 512 + lane * 8 + s] = get(row, ci);
                }
    return dst;
}

template <int RT>
struct PwCfg {
    static constexpr int PG = 8 / RT, PXB = 64 * PG;             // pixel groups of 64 per block, pixels per block
    static constexpr int PDW = 16 * PXB, ADW = RT * 512;          // patch / A dwords per stage
    static constexpr int SDW = PDW + ADW, NST = 6;
    static constexpr int NPD = PDW / (4 * 512);                   // patch DMAs (16 B) per thread per stage: 1 (RT = 4) / 2 (RT = 2)
    static constexpr int NLD = NPD + 1;
    static constexpr int LDS_BYTES = NST * SDW * 4 + 512 * 16;    // + a scratch line per thread for the A DMA of idle threads
};

template <int RT>
__global__ __launch_bounds__(512, 1) void pw_conv_kernel(const PwParams p) {
    using Cfg = PwCfg<RT>;
    constexpr int PG = Cfg::PG, PXB = Cfg::PXB, PDW = Cfg::PDW, SDW = Cfg::SDW, NST = Cfg::NST, NPD = Cfg::NPD, NLD = Cfg::NLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    const int lb = xcd_remap(blockIdx.x, G);
    if (lb >= p.ntasks) return;
    const int mytasks = (p.ntasks - lb + G - 1) / G;
    const int nst = p.nst;
    constexpr unsigned SB = SDW * 4, RING = NST * SB;
    // grouped launch: task -> group (wave-uniform scan of at most a few dozen entries through scalar loads)
    auto group_of = [&](int L) {
        int g = 0;
        while (g + 1 < p.ngroups && L >= p.groups[g + 1].start * p.nrg) ++g;
        return g;
    };
    const unsigned lds0 = (unsigned)(size_t)(wino_lds_void*)smem;

    // ---- issue side: task L -> (row group rg fastest, pixel tile, sample) ----------------------------------------------------------
    unsigned vp[NPD];                       // patch units of this thread: channel u / (PXB / 4), pixels 4 (u % (PXB / 4)) ...
    const bool a_wave = wave < RT * 2;      // waves whose threads copy A units (the others park their DMA in the scratch line)
    int it = lb, is = 0;
    wino_u32x4 d_in, d_a;
    unsigned so_in = 0, so_a = 0;
    int HW = p.HW;                          // pixels per channel plane of the task the issue side serves
    auto issue_task = [&]() {
        int rg, pt, ib;
        const float *tin, *twpk;
        if (p.groups) {
            const int g = group_of(it), local = it - p.groups[g].start * p.nrg;
            rg = local % p.nrg;
            pt = local / p.nrg;
            ib = 0;
            HW = p.groups[g].HW;
            tin = p.groups[g].in;
            twpk = p.groups[g].wpk;
        } else {
            rg = it % p.nrg;
            pt = (it / p.nrg) % p.npt;
            ib = it / (p.nrg * p.npt);
            tin = p.in;
            twpk = p.wpk;
        }
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int u = i * 512 + tid, ch = u / (PXB / 4), q = u - ch * (PXB / 4);
            vp[i] = (unsigned)(ch * HW + pt * PXB + 4 * q) * 4u;
        }
        d_in = wino_rsrc(tin + (long long)ib * (p.in_bs ? p.in_bs : (long long)p.Cin * HW), (unsigned)p.Cin * HW * 4u);
        // the RT row tiles of a group are consecutive images of nst * 2 KB each: thread tid copies unit (tid & 127) of row tile tid >> 7
        const int nrt = (p.Cout + 31) / 32, have = nrt - rg * RT < RT ? nrt - rg * RT : RT;       // (last group may be partial)
        d_a = wino_rsrc(twpk + (long long)rg * RT * nst * 512, (unsigned)have * nst * 2048u);
        so_in = 0;
        so_a = 0;
    };
    issue_task();
    const unsigned va_rt = a_wave ? (unsigned)((tid >> 7) * nst * 2048 + (tid & 127) * 16) : 0x80000000u;
    unsigned islot = lds0;
    auto issue = [&]() {
        const unsigned wb = islot + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < NPD; ++i) wino_dma16(vp[i], d_in, so_in, wb + (unsigned)i * 8192u);
        // A: threads 0 .. RT*128-1 fill the stage's A area in order; the others park their 16 bytes in the scratch line
        const unsigned wa = a_wave ? islot + PDW * 4u + (unsigned)wave * 1024u : lds0 + RING + (unsigned)wave * 1024u;
        wino_dma16(va_rt, d_a, so_a, wa);
        islot = islot + SB == lds0 + RING ? lds0 : islot + SB;
        so_in += 64u * (unsigned)HW;
        so_a += 2048u;
        if (++is == nst) {
            if (it + G < p.ntasks) {
                it += G;
                is = 0;
                issue_task();
            } else {
                is = nst - 1;
                so_in -= 64u * (unsigned)HW;
                so_a -= 2048u;
            }
        }
    };

    // ---- consumer side: wave = row tile (wave / PG) x pixel group (wave % PG) ---------------------------------------------------
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const int rtl = wave / PG, pg = wave % PG;
    const int boff = (lane >> 5) * PXB + pg * 64 + (lane & 31);            // B fragment origin: channel parity plane, pixel
    const int aoff = PDW + rtl * 512 + lane * 8;

#pragma unroll
    for (int j = 0; j < NST - 1; ++j) issue();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (NST - 3)) : "memory");
    __syncthreads();
    unsigned rslot = lds0;
    bool after_epi = false;
    auto frag = [&](unsigned slot) { return reinterpret_cast<const float*>(smem) + (slot - lds0) / 4; };
    float4 a_lo = *reinterpret_cast<const float4*>(frag(lds0) + aoff), a_hi = *reinterpret_cast<const float4*>(frag(lds0) + aoff + 4);
    float b0 = frag(lds0)[boff], b1 = frag(lds0)[boff + 32];
    for (int k = 0, ct = lb; k < mytasks; ++k, ct += G) {
        for (int cs = 0; cs < nst; ++cs) {
            if (after_epi) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD * (NST - 3)) : "memory");
            __syncthreads();
            after_epi = false;
            issue();
            const unsigned nslot = rslot + SB == lds0 + RING ? lds0 : rslot + SB;
            const float* sp = frag(rslot);
            const float* spn = frag(nslot);
            const float av[8] = {a_lo.x, a_lo.y, a_lo.z, a_lo.w, a_hi.x, a_hi.y, a_hi.z, a_hi.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float c0 = b0, c1 = b1;
                if (s < 7) {
                    b0 = sp[boff + (2 * s + 2) * PXB];
                    b1 = sp[boff + (2 * s + 2) * PXB + 32];
                } else {                                                    // the next stage was verified by the barrier above
                    b0 = spn[boff];
                    b1 = spn[boff + 32];
                    a_lo = *reinterpret_cast<const float4*>(spn + aoff);
                    a_hi = *reinterpret_cast<const float4*>(spn + aoff + 4);
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], c0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], c1, acc[1], 0, 0, 0);
            }
            rslot = nslot;
        }
        // ---- epilogue: 32 rows x 64 pixels per wave, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), pixel lane & 31 -------------------
        {
            int rg, pt, b, eHW;
            float* tout;
            if (p.groups) {
                const int g = group_of(ct), local = ct - p.groups[g].start * p.nrg;
                rg = local % p.nrg;
                pt = local / p.nrg;
                b = 0;
                eHW = p.groups[g].HW;
                tout = p.groups[g].out;
            } else {
                rg = ct % p.nrg;
                pt = (ct / p.nrg) % p.npt;
                b = ct / (p.nrg * p.npt);
                eHW = p.HW;
                tout = p.out;
            }
            const int row0 = (rg * RT + rtl) * 32 + 4 * (lane >> 5);
            float* ob = tout + (long long)b * p.Cout * eHW + pt * PXB + pg * 64 + (lane & 31);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + (r & 3) + 8 * (r >> 2);
                    if (row < p.Cout) ob[(long long)row * eHW + n * 32] = acc[n][r];
                    acc[n][r] = 0.f;
                }
            after_epi = true;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no LDS-DMA may outlive the block
}

inline bool pw_supported(int Cin, int Cout, int HW) { return Cin % 16 == 0 && HW % 256 == 0 && Cout >= 32; }
hipError_t conv_pw(PwParams p, hipStream_t s);              // conv_inst_wino.hip
// grouped launch: p.groups / p.ngroups (device array), p.Cin, p.Cout; ntasks = (sum of the groups' pixel tiles) x row groups
inline int pw_group_tasks(int Cout, int HW) { return ((((Cout + 31) / 32) + 3) / 4) * (HW / 128); }
hipError_t conv_pw_grouped(PwParams p, int ntasks, hipStream_t s);

}  // namespace chk
