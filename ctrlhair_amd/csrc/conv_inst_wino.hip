// conv_inst_wino.hip -- instantiations + launchers of the Winograd F(2x2,3x3) exact-f32 MFMA kernels (conv_wino.h)
#include "conv_pw.h"
#include <algorithm>
#include "conv_wino.h"

namespace chk {

static int wino_num_cus() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return cus[dev];
}

template <class K>
static hipError_t wino_attr(K kern, int bytes, bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        done[dev] = true;
    }
    return hipSuccess;
}

// out = act(sum over the K slices (in slice order) + bias + residual): the second pass of a split-K launch of wino_plain_kernel
__global__ __launch_bounds__(256) void wino_splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, const float* __restrict__ bias,
                                                                 const float* __restrict__ res, int res_up, int act, int ks, int B, int C, int H, int W) {
    const long long n4 = (long long)B * C * H * W / 4, slab = n4 * 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4*>(partial)[i];
        for (int k = 1; k < ks; ++k) {
            const float4 t = reinterpret_cast<const float4*>(partial + k * slab)[i];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const long long e = 4 * i;
        const int x = (int)(e % W), y = (int)((e / W) % H);
        const long long bc = e / ((long long)W * H);
        const float bv = bias ? bias[bc % C] : 0.f;
        a.x += bv; a.y += bv; a.z += bv; a.w += bv;
        if (res) {
            if (res_up) {
                const float* rp = res + bc * (long long)(H >> 1) * (W >> 1) + (long long)(y >> 1) * (W >> 1) + (x >> 1);
                a.x += rp[0]; a.y += rp[0]; a.z += rp[1]; a.w += rp[1];
            } else {
                const float4 r = reinterpret_cast<const float4*>(res)[i];
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
        }
        if (act != ACT_NONE) { a.x = apply_act(a.x, act); a.y = apply_act(a.y, act); a.z = apply_act(a.z, act); a.w = apply_act(a.w, act); }
        reinterpret_cast<float4*>(out)[i] = a;
    }
}

hipError_t conv_wino_plain(WinoParams p, hipStream_t s) {
    if ((!wino_supported(p.H, p.W, p.Cin) && !(wino_supported_pair16(p.B, p.H, p.W, p.Cin) && !p.reflect && !p.res_up)) || !p.zero) return hipErrorInvalidValue;
    if (p.d2s && (p.Cout % 4 || p.res || p.reflect || p.in_up || !wino_supported(p.H, p.W, p.Cin))) return hipErrorInvalidValue;
    if (p.in_up && (p.reflect || !wino_supported(p.H, p.W, p.Cin))) return hipErrorInvalidValue;      // (H, W: the conv's own = 2 x stored size)
    wino_fill_launch(p);
    p.ksplit = 1;
    if (p.partial && !p.d2s && p.W % 4 == 0) {        // far fewer tasks than CUs and a long k-loop: slices of the input channels
        const int cus = wino_num_cus();
        int ks = 1;
        while (ks < 8 && 2 * ks * p.ntasks <= cus && p.nks % (2 * ks) == 0 && p.nks / (2 * ks) >= 16 &&
               (long long)2 * ks * p.B * p.Cout * p.H * p.W <= p.partial_cap)
            ks *= 2;
        if (ks > 1) {
            p.ksplit = ks;
            p.ntasks *= ks;
        }
    }
    if (p.nks / p.ksplit < wino::NST) p.claim = nullptr;     // (the issue side may run two tasks ahead: static split only)
    const int grid = p.ntasks < wino_num_cus() ? p.ntasks : wino_num_cus();
    static bool d0[64] = {}, d1[64] = {}, d2[64] = {};
    if (p.d2s) {
        hipError_t e = wino_attr(wino_plain_kernel<1>, wino::LDS_BYTES, d1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_plain_kernel<1>, dim3(grid), dim3(512), wino::LDS_BYTES, s, p);
        return hipGetLastError();
    }
    if (p.in_up) {
        hipError_t e = wino_attr(wino_plain_kernel<2>, wino::LDS_BYTES, d2);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_plain_kernel<2>, dim3(grid), dim3(512), wino::LDS_BYTES, s, p);
    } else {
        hipError_t e = wino_attr(wino_plain_kernel<0>, wino::LDS_BYTES, d0);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_plain_kernel<0>, dim3(grid), dim3(512), wino::LDS_BYTES, s, p);
    }
    if (p.ksplit > 1) {
        const long long n4 = (long long)p.B * p.Cout * p.H * p.W / 4;
        const int blocks = (int)std::min<long long>((n4 + 255) / 256, 4096);
        hipLaunchKernelGGL(wino_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.partial, p.out, p.bias, p.res, p.res_up, p.act, p.ksplit, p.B, p.Cout, p.H,
                           p.W);
    }
    return hipGetLastError();
}


template <int RT>
static hipError_t launch_pw(PwParams p, hipStream_t s) {
    using Cfg = PwCfg<RT>;
    static bool d0[64] = {};
    hipError_t e = wino_attr(pw_conv_kernel<RT>, Cfg::LDS_BYTES, d0);
    if (e != hipSuccess) return e;
    const int nrt = (p.Cout + 31) / 32;
    p.nrg = (nrt + RT - 1) / RT;
    p.npt = p.HW / Cfg::PXB;
    p.ntasks = p.B * p.npt * p.nrg;
    p.nst = p.Cin / 16;
    const int grid = p.ntasks < wino_num_cus() ? p.ntasks : wino_num_cus();
    hipLaunchKernelGGL(pw_conv_kernel<RT>, dim3(grid), dim3(512), Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}
hipError_t conv_pw(PwParams p, hipStream_t s) {
    if (!pw_supported(p.Cin, p.Cout, p.HW)) return hipErrorInvalidValue;
    p.groups = nullptr;
    p.ngroups = 0;
    return (p.Cout + 31) / 32 >= 4 ? launch_pw<4>(p, s) : launch_pw<2>(p, s);
}
// several GEMMs (same Cin, same Cout, own operands and pixel counts) as ONE persistent launch of the RT = 4 kernel (conv_pw.h)
hipError_t conv_pw_grouped(PwParams p, int ntasks, hipStream_t s) {
    using Cfg = PwCfg<4>;
    if (p.Cin % 16 || p.Cout < 1 || !p.groups || p.ngroups < 1 || ntasks < 1) return hipErrorInvalidValue;      // (rows beyond Cout: zero rows of the packed operand, masked at the store)
    static bool d0[64] = {};
    hipError_t e = wino_attr(pw_conv_kernel<4>, Cfg::LDS_BYTES, d0);
    if (e != hipSuccess) return e;
    p.in = p.wpk = nullptr;
    p.out = nullptr;
    p.B = 1;
    p.HW = 0;
    p.nrg = ((p.Cout + 31) / 32 + 3) / 4;
    p.npt = 0;
    p.ntasks = ntasks;
    p.nst = p.Cin / 16;
    const int grid = ntasks < wino_num_cus() ? ntasks : wino_num_cus();
    hipLaunchKernelGGL(pw_conv_kernel<4>, dim3(grid), dim3(512), Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}

template <int TH>
static hipError_t launch_wino_ace(const WinoAceParams& p, hipStream_t s) {
    static bool d0[64] = {};
    hipError_t e = wino_attr(wino_ace_kernel<TH>, WaCfg<TH>::LDS_BYTES, d0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wino_ace_kernel<TH>, dim3(wino_num_cus()), dim3(512), WaCfg<TH>::LDS_BYTES, s, p);
    return hipGetLastError();
}
hipError_t conv_wino_ace(WinoAceParams p, hipStream_t s) {
    if (p.gq) {            // gather mode
        if (!p.gq_n || p.gq_cap <= 0 || p.B > 32 || !p.work || !p.total || (p.C & 3) || (p.W & 1) || (p.H & 1)) return hipErrorInvalidValue;
        p.nrt = (p.C + 15) / 16;
        if (p.nrt > 2 * 65535) return hipErrorInvalidValue;
        p.K = 128 + (p.wsty ? 20 : 0);
        static bool d0[64] = {};
        hipError_t e = wino_attr(wino_ace_gather_kernel<0>, winog::LDS_BYTES, d0);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_ace_gather_kernel<0>, dim3(wino_num_cus()), dim3(512), winog::LDS_BYTES, s, p);
        return hipGetLastError();
    }
    if ((p.TH != 16 && p.TH != 32) || p.H % p.TH || p.W % wino::TW || !p.zero || !p.qlist || !p.qcnt || !p.work || !p.total || (p.C & 3))
        return hipErrorInvalidValue;
    p.nrt = (p.C + 15) / 16;
    p.ntx = p.W / wino::TW;
    p.nty = p.H / p.TH;
    p.K = 128 + (p.wsty ? 20 : 0);
    return p.TH == 32 ? launch_wino_ace<32>(p, s) : launch_wino_ace<16>(p, s);
}

// ---- gather mode: per-tile lists (tiles of 32 x 16) -> one list of boundary quads per sample, tasks of 64 consecutive entries --------
__global__ __launch_bounds__(1024) void wino_gather_scan_kernel(const int* __restrict__ qcnt, int* __restrict__ qoff, int* __restrict__ gq_n, int tps) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < tps; t0 += 1024) {
        const int t = t0 + tid;
        const int c = t < tps ? qcnt[b * tps + t] : 0;
        int v = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64);
            if (lane >= off) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        if (t < tps) qoff[b * tps + t] = base + v - c;
        __syncthreads();
        if (tid == 1023) carry = base + v;
        __syncthreads();
    }
    if (tid == 0) gq_n[b] = carry;
}
__global__ __launch_bounds__(128) void wino_gather_fill_kernel(const uint8_t* __restrict__ qlist, const int* __restrict__ qcnt, const int* __restrict__ qoff,
                                                               unsigned* __restrict__ gq, int gq_cap, int ntx, int nty) {
    const int tile = blockIdx.x, i = threadIdx.x;
    if (i >= qcnt[tile]) return;
    const int b = tile / (ntx * nty), tr = tile % (ntx * nty);
    const int q = qlist[(long long)tile * 128 + i];
    const unsigned y = (unsigned)((tr / ntx) * 16 + 2 * (q >> 4)), x = (unsigned)((tr % ntx) * 32 + 2 * (q & 15));
    gq[(long long)b * gq_cap + qoff[tile] + i] = y << 16 | x;
}
hipError_t wino_gather_lists(const uint8_t* qlist, const int* qcnt, int* qoff, unsigned* gq, int* gq_n, int gq_cap, int B, int H, int W, hipStream_t s) {
    if (H % 16 || W % 32 || B > 32 || H > 65535 || W > 65535) return hipErrorInvalidValue;
    const int ntx = W / 32, nty = H / 16;
    hipLaunchKernelGGL(wino_gather_scan_kernel, dim3(B), dim3(1024), 0, s, qcnt, qoff, gq_n, ntx * nty);
    hipLaunchKernelGGL(wino_gather_fill_kernel, dim3(B * ntx * nty), dim3(128), 0, s, qlist, qcnt, qoff, gq, gq_cap, ntx, nty);
    return hipGetLastError();
}
__global__ __launch_bounds__(1024) void wino_gather_worklist_kernel(const int* __restrict__ gq_n, const int* __restrict__ pcnt, int B, int tps, int nrt,
                                                                    unsigned* __restrict__ work, int* __restrict__ total) {
    __shared__ int base[33];
    __shared__ int stat[4];
    const int tid = threadIdx.x;
    const int npair = (nrt + 1) >> 1;
    if (tid < 4) stat[tid] = 0;
    if (tid == 0) {
        int o = 0, sq = 0, sg = 0;
        for (int b = 0; b < B; ++b) {
            base[b] = o;
            const int nq = gq_n[b];
            o += ((nq + 63) >> 6) * npair;
            sq += nq;
            sg += (nq + 15) >> 4;
        }
        base[B] = o;
        stat[0] = sq;
        stat[1] = sg;
    }
    __syncthreads();
    int sp = 0;
    for (int t = tid; t < B * tps; t += 1024) sp += pcnt[t];
    atomicAdd(&stat[2], sp);
    for (int b = 0; b < B; ++b) {
        const int ne = base[b + 1] - base[b];
        for (int e = tid; e < ne; e += 1024) work[base[b] + e] = (unsigned)b | (unsigned)(e / npair) << 5 | (unsigned)(e % npair) << 16;
    }
    __syncthreads();
    if (tid == 0) {
        total[0] = total[4] = base[B];
        total[1] = stat[0];
        total[5] = stat[2];
        total[2] = total[6] = stat[1];
        total[3] = total[7] = stat[1] * nrt;
    }
}
hipError_t wino_gather_worklist(const int* gq_n, const int* pcnt, int B, int tiles_per_sample, int nrt, unsigned* work, int* total, hipStream_t s) {
    if (B > 32 || nrt > 2 * 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wino_gather_worklist_kernel, dim3(1), dim3(1024), 0, s, gq_n, pcnt, B, tiles_per_sample, nrt, work, total);
    return hipGetLastError();
}

// ---- boundary quads of a tile of 32 x TH pixels: one block of 8 TH threads = the tile's 16 x TH / 2 quads, ordered compaction -----
template <int TH>
__global__ __launch_bounds__(8 * TH) void wino_quad_list_kernel(const uint8_t* __restrict__ u5, uint8_t* __restrict__ qlist, int* __restrict__ qcnt,
                                                                int* __restrict__ pcnt, int H, int W, int ntx, int nty) {
    constexpr int NW = TH / 8;
    __shared__ int wc[NW], pc[NW];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int b = tile / (ntx * nty), tr = tile % (ntx * nty);
    const int y = (tr / ntx) * TH + 2 * (tid >> 4), x = (tr % ntx) * 32 + 2 * (tid & 15);
    int npx = 4;                                               // boundary pixels of the quad (u5 == nullptr: every pixel is one)
    if (u5) {
        const uint8_t* up = u5 + ((long long)b * H + y) * W + x;
        npx = (up[0] == 255) + (up[1] == 255) + (up[W] == 255) + (up[W + 1] == 255);
    }
    const bool bnd = npx > 0;
    const unsigned long long m = __ballot(bnd);
    const int lane = tid & 63, wave = tid >> 6;
    int ps = npx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off, 64);
    if (lane == 0) { wc[wave] = __popcll(m); pc[wave] = ps; }
    __syncthreads();
    int base = 0, tq = 0, tp = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) base += wc[w];
        tq += wc[w];
        tp += pc[w];
    }
    if (bnd) qlist[(long long)tile * (8 * TH) + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)tid;
    if (tid == 0) { qcnt[tile] = tq; pcnt[tile] = tp; }
}
hipError_t wino_quad_lists(const uint8_t* u5, uint8_t* qlist, int* qcnt, int* pcnt, int B, int H, int W, int TH, hipStream_t s) {
    if ((TH != 16 && TH != 32) || H % TH || W % 32) return hipErrorInvalidValue;
    const int ntx = W / 32, nty = H / TH;
    if (TH == 32) hipLaunchKernelGGL(wino_quad_list_kernel<32>, dim3(B * ntx * nty), dim3(256), 0, s, u5, qlist, qcnt, pcnt, H, W, ntx, nty);
    else hipLaunchKernelGGL(wino_quad_list_kernel<16>, dim3(B * ntx * nty), dim3(128), 0, s, u5, qlist, qcnt, pcnt, H, W, ntx, nty);
    return hipGetLastError();
}

// ---- block tasks of conv_wino_ace: per tile ceil(quads / 64) parts x ceil(nrt / 2) row pairs; one block of 1024 threads scans -------
// total: [0] entries, [1] boundary quads, [2] 16-quad groups, [3] groups x row tiles (wave tasks: x 32 rows x 16 quads of accumulators),
// [4..7] the same block with the boundary PIXELS in place of the quads (the layout the profiling records of ace_sparse.h expect)
__global__ __launch_bounds__(1024) void wino_ace_worklist_kernel(const int* __restrict__ qcnt, const int* __restrict__ pcnt, int ntiles, int nrt,
                                                                 unsigned* __restrict__ work, int* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    __shared__ int stat[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int npair = (nrt + 1) >> 1;
    if (tid == 0) carry = 0;
    if (tid < 4) stat[tid] = 0;
    __syncthreads();
    int s_q = 0, s_g = 0, s_w = 0, s_p = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int tile = t0 + tid;
        int ne = 0, halves = 0;
        if (tile < ntiles) {
            const int c = qcnt[tile], g = (c + 15) >> 4;
            halves = (c + 63) >> 6;
            ne = halves * npair;
            s_q += c;
            s_p += pcnt[tile];
            s_g += g;
            s_w += g * nrt;
        }
        int v = ne;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64);
            if (lane >= off) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int o = base + v - ne;
        for (int h = 0; h < halves; ++h)
            for (int pr = 0; pr < npair; ++pr) work[o++] = (unsigned)tile | ((unsigned)pr << 20) | ((unsigned)h << 30);
        __syncthreads();
        if (tid == 1023) carry = base + v;
        __syncthreads();
    }
    atomicAdd(&stat[0], s_q);
    atomicAdd(&stat[1], s_g);
    atomicAdd(&stat[2], s_w);
    atomicAdd(&stat[3], s_p);
    __syncthreads();
    if (tid == 0) {
        total[0] = total[4] = carry;
        total[1] = stat[0];
        total[5] = stat[3];
        total[2] = total[6] = stat[1];
        total[3] = total[7] = stat[2];
    }
}
hipError_t wino_ace_worklist(const int* qcnt, const int* pcnt, int ntiles, int nrt, unsigned* work, int* total, hipStream_t s) {
    if (ntiles >= (1 << 20) || nrt >= (1 << 11)) return hipErrorInvalidValue;      // (pairs < 2^10, parts < 4)
    hipLaunchKernelGGL(wino_ace_worklist_kernel, dim3(1), dim3(1024), 0, s, qcnt, pcnt, ntiles, nrt, work, total);
    return hipGetLastError();
}

// ---- style LUT -> per-sample Winograd A images --------------------------------------------------------------------------------------
// lut[((b*19 + j)*9 + t)*2C + gb*C + c]  (P = W[:, :, t] relu(fc_mu_j(code)), blend factor folded in; sean_model.cpp)
// wsty[((b*nrt + rt)*5 + s)*2048 + (idx*64 + lane)*4 + e]: fragment a = 4 idx + e = (xi = a >> 1, m = a & 1) of row (m ? beta : gamma) of
// channel 16 rt + (lane & 15), input "channel" = label j = 4 s + (lane >> 4) (j = 19: the zero plane).  U = G P G^T in f32.
__global__ __launch_bounds__(256) void wino_style_pack_kernel(const float* __restrict__ lut, float* __restrict__ wsty, int B, int C, int nrt) {
    const long long n = (long long)B * nrt * 5 * 512;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), idx = (int)((i >> 6) & 7), s = (int)((i >> 9) % 5);
    const int rt = (int)((i / (512 * 5)) % nrt), b = (int)(i / (512LL * 5 * nrt));
    const int c = rt * 16 + (lane & 15), j = 4 * s + (lane >> 4);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C && j < 19) {
        const float* P = lut + ((long long)(b * 19 + j) * 9) * 2 * C + c;
        float u[2][2];
#pragma unroll
        for (int gb = 0; gb < 2; ++gb) {
            float g[3][3];
#pragma unroll
            for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = P[(long long)t * 2 * C + gb * C];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int xi = 2 * idx + e, ii = xi >> 2, jj = xi & 3;
                // row ii of G applied to the rows of g, then column jj:  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
                float r[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    r[q] = ii == 0 ? g[0][q] : (ii == 3 ? g[2][q] : 0.5f * (g[0][q] + (ii == 1 ? g[1][q] : -g[1][q]) + g[2][q]));
                u[e][gb] = jj == 0 ? r[0] : (jj == 3 ? r[2] : 0.5f * (r[0] + (jj == 1 ? r[1] : -r[1]) + r[2]));
            }
        }
        o = make_float4(u[0][0], u[0][1], u[1][0], u[1][1]);
    }
    reinterpret_cast<float4*>(wsty)[i] = o;
}
hipError_t wino_style_pack(const float* lut, float* wsty, int B, int C, hipStream_t s) {
    const int nrt = (C + 15) / 16;
    const long long n = (long long)B * nrt * 5 * 512;
    hipLaunchKernelGGL(wino_style_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lut, wsty, B, C, nrt);
    return hipGetLastError();
}

}  // namespace chk
