// ace_sparse.hip -- label classification, work lists, per-label gamma/beta tables and the elementwise interior pass of the
// exact SPADE-interior reduction (ace_sparse.h; reference: normalization.py:108-189,249-257).
#include "ace_sparse.h"

#include "conv_mfma.h"
#include "sh16.h"

namespace chk {

// ---- classification: one block = one tile of 32 x TH pixels of one sample (TH * 32 threads) -------------------------------
// A pixel is INTERIOR iff its label is < 19 and all 25 labels of its 5x5 neighbourhood exist (inside the image) and equal it.
// need[p] = 1 iff some pixel of the 3x3 neighbourhood of p is a boundary pixel: only there does the boundary conv read the
// SPADE hidden activations, so the label-table kernel may skip every other pixel.
// e16 != nullptr: a pixel that is not interior but whose 5x5 neighbourhood (inside the image, labels < 19) is five uniform columns
// A^s B^(5-s) or five uniform rows likewise (s = 1..4, A != B) is a STRAIGHT-EDGE pixel: u5 = 253, e16 = its code (ace_sparse.h); it is
// not a boundary pixel for `need`, `list` and `cnt`.
__device__ __forceinline__ int ace_edge_code(const uint8_t* w, int PW) {      // w: top-left of the 5x5 window in the LDS patch; -1: none
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int sj = o ? PW : 1, si = o ? 1 : PW;          // step along the split direction / along a line of equal labels
        int l[5];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            l[j] = w[j * sj];
#pragma unroll
            for (int i = 1; i < 5; ++i) ok = ok && w[j * sj + i * si] == l[j];
        }
        const int A = l[0], Bl = l[4];
        ok = ok && A < 19 && Bl < 19 && A != Bl;
        int s = 1;
#pragma unroll
        for (int j = 1; j < 4; ++j) s += (l[j] == A && s == j) ? 1 : 0;       // leading run of A
#pragma unroll
        for (int j = 1; j < 4; ++j) ok = ok && l[j] == (j < s ? A : Bl);
        if (ok) return ((o * 19 + A) * 19 + Bl) * 4 + (s - 1);
    }
    return -1;
}

template <int TH>
__global__ __launch_bounds__(32 * TH) void ace_classify_kernel(const uint8_t* __restrict__ lab, uint8_t* __restrict__ u5,
                                                             uint8_t* __restrict__ need, uint16_t* __restrict__ list,
                                                             int* __restrict__ cnt, int H, int W, int tiles_x, int tiles_y,
                                                             uint16_t* __restrict__ e16) {
    constexpr int NT = 32 * TH, PW = 38, PH = TH + 6, BW = 34, BH = TH + 2, NW = NT / 64;
    __shared__ uint8_t patch[PH * PW];       // labels of the tile + 3 pixels around it (255 outside the image)
    __shared__ uint8_t bflag[BH * BW];       // boundary flags of the tile + 1 pixel around it
    __shared__ uint16_t ecode[BH * BW];      // codes of the straight-edge pixels among them
    __shared__ int wcnt[NW];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * 32;
    const uint8_t* lb = lab + (long long)b * H * W;
    for (int i = tid; i < PH * PW; i += NT) {
        const int y = y0 - 3 + i / PW, x = x0 - 3 + i % PW;
        patch[i] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? lb[(long long)y * W + x] : (uint8_t)255;
    }
    __syncthreads();
    for (int i = tid; i < BH * BW; i += NT) {
        const int by = i / BW, bx = i % BW;                       // pixel (y0 - 1 + by, x0 - 1 + bx)
        const int y = y0 - 1 + by, x = x0 - 1 + bx;
        const uint8_t c = patch[(by + 2) * PW + bx + 2];
        bool uni = c < 19;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) uni = uni && patch[(by + dy) * PW + bx + dx] == c;
        const bool inside = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        uint8_t f = inside ? (uni ? c : (uint8_t)255) : (uint8_t)254;       // 254: outside the image (neither kind)
        if (e16 && f == 255) {
            const int code = ace_edge_code(patch + by * PW + bx, PW);
            if (code >= 0) {
                f = (uint8_t)ACE_EDGE;
                ecode[i] = (uint16_t)code;
            }
        }
        bflag[i] = f;
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31, y = y0 + ty, x = x0 + tx;
    const bool inside = y < H && x < W;
    const uint8_t me = bflag[(ty + 1) * BW + tx + 1];
    bool nd = false;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) nd = nd || bflag[(ty + dy) * BW + tx + dx] == 255;
    if (inside) {
        u5[(long long)b * H * W + (long long)y * W + x] = me;
        need[(long long)b * H * W + (long long)y * W + x] = nd ? 1 : 0;
        if (e16 && me == ACE_EDGE) e16[(long long)b * H * W + (long long)y * W + x] = ecode[(ty + 1) * BW + tx + 1];
    }
    const bool bnd = inside && me == 255;
    // ordered compaction: raster order inside the tile (wave w = rows 2w, 2w+1)
    const unsigned long long m = __ballot(bnd);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w < wave) base += wcnt[w];
        tot += wcnt[w];
    }
    if (bnd) list[(long long)tile * NT + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    if (tid == 0) cnt[tile] = tot;
}

hipError_t ace_classify(const uint8_t* lab, uint8_t* u5, uint8_t* need, uint16_t* list, int* cnt, int B, int H, int W, int TH,
                        hipStream_t s, uint16_t* e16) {
    const int tx = (W + 31) / 32, ty = (H + TH - 1) / TH;
    if (TH == 8) hipLaunchKernelGGL(ace_classify_kernel<8>, dim3(B * tx * ty), dim3(256), 0, s, lab, u5, need, list, cnt, H, W, tx, ty, e16);
    else if (TH == 16) hipLaunchKernelGGL(ace_classify_kernel<16>, dim3(B * tx * ty), dim3(512), 0, s, lab, u5, need, list, cnt, H, W, tx, ty, e16);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- work list: exclusive scan of the block tasks per tile, one block of 1024 threads ------------------------------------
// mode 0: block tasks of conv_ace_sparse_kernel (entry = tile | block task << 20); mode 1: tile-skip mode of the f16x3
// wave-specialised kernel -- one entry (tile | row tile << 20) per row tile of every spatial tile with a boundary pixel, the
// tile's 512 pixels all go through the conv (statistics count them as such); mode 2: the same entries for the compacting
// variant of that kernel (statistics count the boundary pixels / their 32-pixel sub-tiles); mode 3: as 2, but a spatial tile
// with at most four sub-tiles gets one entry per PAIR of row tiles, in the second list `work2` (row-tile field = pair index:
// conv_sh16_ws_kernel<..., CP = 2>)
__global__ __launch_bounds__(1024) void ace_worklist_kernel(const int* __restrict__ cnt, int ntiles, int mtiles, int mode, int tile_px,
                                                            unsigned* __restrict__ work, int* __restrict__ total,
                                                            unsigned* __restrict__ work2, int* __restrict__ total2,
                                                            unsigned* __restrict__ work3) {
    __shared__ int wsum[16], wsum2[16], wsum3[16];
    __shared__ int carry, carry2, carry3;
    __shared__ int stat[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = carry2 = carry3 = 0;
    if (tid < 3) stat[tid] = 0;
    __syncthreads();
    int s_px = 0, s_sub = 0, s_ws = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int tile = t0 + tid;
        int nbt = 0, nbp = 0, nbq = 0;                 // entries of this tile in `work` / `work2` (pairs) / `work3` (quads), mode 3
        if (tile < ntiles) {
            int c = cnt[tile];
            if (mode == 1 && c > 0) c = tile_px;
            const int NS = (c + 31) >> 5;
            if (mode == 3 && work3 && NS >= 1 && NS <= 2) {   // one entry per four row tiles (conv_sh16_ws_kernel<..., CP = 3>)
                nbq = (mtiles + 3) >> 2;
            } else if (mode == 3 && NS >= 1 && NS <= 4) {     // one entry per pair of row tiles (CP = 2)
                nbp = (mtiles + 1) >> 1;
            } else if (mode >= 1) {    // mode 2 / 3: the same entries, but the conv only runs over the compacted boundary pixels
                nbt = c > 0 ? mtiles : 0;
            } else {
                int ng, per;
                sparse_groups(NS, mtiles, ng, per);
                nbt = (ng * mtiles + 3) >> 2;
            }
            s_px += c;
            s_sub += NS;
            s_ws += NS * mtiles;
        }
        int v = nbt, vp = nbp, vq = nbq;               // inclusive scans inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64), up = __shfl_up(vp, off, 64), uq = __shfl_up(vq, off, 64);
            if (lane >= off) { v += u; vp += up; vq += uq; }
        }
        if (lane == 63) { wsum[wave] = v; wsum2[wave] = vp; wsum3[wave] = vq; }
        __syncthreads();
        int base = carry, base2 = carry2, base3 = carry3;
        for (int w = 0; w < wave; ++w) { base += wsum[w]; base2 += wsum2[w]; base3 += wsum3[w]; }
        const int excl = base + v - nbt, excl2 = base2 + vp - nbp, excl3 = base3 + vq - nbq;
        for (int i = 0; i < nbt; ++i) work[excl + i] = (unsigned)tile | ((unsigned)i << 20);
        for (int i = 0; i < nbp; ++i) work2[excl2 + i] = (unsigned)tile | ((unsigned)i << 20);
        for (int i = 0; i < nbq; ++i) work3[excl3 + i] = (unsigned)tile | ((unsigned)i << 20);
        __syncthreads();
        if (tid == 1023) { carry = base + v; carry2 = base2 + vp; carry3 = base3 + vq; }
        __syncthreads();
    }
    atomicAdd(&stat[0], s_px);
    atomicAdd(&stat[1], s_sub);
    atomicAdd(&stat[2], s_ws);
    __syncthreads();
    if (tid == 0) {
        total[0] = carry;
        total[1] = stat[0];
        total[2] = stat[1];
        total[3] = stat[2];
        if (total2) { total2[0] = carry2; total2[1] = carry3; }
    }
}

hipError_t ace_worklist(const int* cnt, int ntiles, int mtiles, unsigned* work, int* total, hipStream_t s, int mode, int tile_px,
                        unsigned* work2, int* total2, unsigned* work3) {
    if (ntiles >= (1 << 20) || mtiles >= (1 << 12)) return hipErrorInvalidValue;
    if (mode == 3 && !(work2 && total2)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ace_worklist_kernel, dim3(1), dim3(1024), 0, s, cnt, ntiles, mtiles, mode, tile_px, work, total, work2, total2, work3);
    return hipGetLastError();
}

// ---- per-(sample, label) gamma/beta rows of the interior pixels ------------------------------------------------------------
__global__ __launch_bounds__(256) void ace_gtable_kernel(const float* __restrict__ bias_g, const float* __restrict__ bias_b,
                                                         const float* __restrict__ gconst, const float* __restrict__ lut,
                                                         int lut_rs, int lut_ns, int lut_bs, float lut_mul,
                                                         float* __restrict__ gtab, int B, int C) {
    const long long n = (long long)B * 19 * 2 * C;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C), gb = (int)((i / C) & 1), j = (int)((i / (2 * C)) % 19), b = (int)(i / (2LL * C * 19));
    float v = (gb ? bias_b : bias_g)[c] + gconst[((long long)j * 2 + gb) * C + c];
    if (lut) {
        float sacc = 0.f;
        const long long col = (long long)(b * lut_bs + j) * lut_ns;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const long long row = (long long)(t * 2 + gb) * C + c;
            // lut_rs == 1: rows contiguous per column; else the C4 layout [row / 4][column][4] written by the f16x3 LUT GEMM
            sacc += lut_rs == 1 ? lut[row + col] : lut[(row >> 2) * 4 * lut_rs + col + (row & 3)];
        }
        v += sacc * lut_mul;
    }
    gtab[i] = v;
}

hipError_t ace_gtable(const float* bias_g, const float* bias_b, const float* gconst, const float* lut, int lut_rs, int lut_ns,
                      int lut_bs, float lut_mul, float* gtab, int B, int C, hipStream_t s) {
    const long long n = (long long)B * 19 * 2 * C;
    hipLaunchKernelGGL(ace_gtable_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bias_g, bias_b, gconst, lut, lut_rs,
                       lut_ns, lut_bs, lut_mul, gtab, B, C);
    return hipGetLastError();
}

// ---- straight-edge pixels: the per-code rows of an ACE (ch_finalize) and the per-call column / row sums of the style LUT ----------------
// hidden-vector index of the window (X, Y, Z) of three column (row) labels out of {A, B}, monotone: XXX -> a_X; AAB / ABB -> pair entries
__host__ __device__ inline int ace_edge_hv(int X, int Y, int Z) { return (X == Z) ? X : 19 + (X * 19 + Z) * 2 + (Y == X ? 0 : 1); }
// labels of the five columns (rows) of code (A, B, s): A for index < s, B from s on; hidden position d = -1, 0, 1 sees columns 1 + d .. 3 + d
__global__ __launch_bounds__(256) void ace_edge_table_kernel(const double* __restrict__ W6, const double* __restrict__ hv, const float* __restrict__ bias_g,
                                                             const float* __restrict__ bias_b, float scale_g, float scale_b, float* __restrict__ E, int C) {
    const long long n = (long long)ACE_EDGE_CODES * 2 * C;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C), gb = (int)((i / C) & 1), code = (int)(i / (2LL * C));
    const int s = (code & 3) + 1, Bl = (code >> 2) % 19, A = ((code >> 2) / 19) % 19, o = (code >> 2) / 361;
    double acc = 0.0;
    if (A != Bl) {
        const double* w = W6 + (long long)gb * 128 * 6 * C + (long long)(o * 3) * C + c;      // [gb][k][6][C]: consecutive threads, consecutive c
#pragma unroll
        for (int d = 0; d < 3; ++d) {                            // hidden position d - 1: window = line labels d, d + 1, d + 2
            const int X = d < s ? A : Bl, Y = d + 1 < s ? A : Bl, Z = d + 2 < s ? A : Bl;
            const double* h = hv + ((long long)o * 741 + ace_edge_hv(X, Y, Z)) * 128;
            double a = 0.0;
            for (int k = 0; k < 128; ++k) a += w[((long long)k * 6 + d) * C] * h[k];
            acc += a;
        }
    }
    E[i] = (float)(acc * (gb ? scale_b : scale_g) + (double)(gb ? bias_b : bias_g)[c]);
}
hipError_t ace_edge_table(const double* W6, const double* hv, const float* bias_g, const float* bias_b, float scale_g, float scale_b, float* E, int C,
                          hipStream_t s) {
    const long long n = (long long)ACE_EDGE_CODES * 2 * C;
    hipLaunchKernelGGL(ace_edge_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W6, hv, bias_g, bias_b, scale_g, scale_b, E, C);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void ace_p6table_kernel(const float* __restrict__ lut, int lut_rs, int lut_ns, int lut_bs, float lut_mul,
                                                          float* __restrict__ p6, int B, int C) {
    const long long n = (long long)B * 19 * 6 * 2 * C;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C), gb = (int)((i / C) & 1), k = (int)((i / (2 * C)) % 6), j = (int)((i / (12LL * C)) % 19), b = (int)(i / (12LL * C * 19));
    const long long col = (long long)(b * lut_bs + j) * lut_ns;
    float sacc = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = k < 3 ? u * 3 + k : (k - 3) * 3 + u;        // column k: taps (dy = u - 1, dx = k - 1); row k - 3: taps (dy = k - 4, dx = u - 1)
        const long long row = (long long)(t * 2 + gb) * C + c;
        sacc += lut_rs == 1 ? lut[row + col] : lut[(row >> 2) * 4 * lut_rs + col + (row & 3)];
    }
    p6[i] = sacc * lut_mul;
}
hipError_t ace_p6table(const float* lut, int lut_rs, int lut_ns, int lut_bs, float lut_mul, float* p6, int B, int C, hipStream_t s) {
    const long long n = (long long)B * 19 * 6 * 2 * C;
    hipLaunchKernelGGL(ace_p6table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lut, lut_rs, lut_ns, lut_bs, lut_mul, p6, B, C);
    return hipGetLastError();
}

// ---- interior pass, exact-f32 path (NCHW in, NCHW out): HBM-bound streaming ----------------------------------------------
// One thread = 4 consecutive pixels of a row (16-byte loads / stores when all four are interior), one block = 1024 pixels of
// one sample x CG channels; the sample's table slice [19][2][CG] sits in LDS (row pitch padded by one float: lanes that hold
// different labels hit different banks).
//      out = act((bn_a x + nv nz + bn_d) (1 + gamma) + beta)          (normalization.py:111-112,182; architecture.py:95)
constexpr int IN_CG = 32;
#ifdef CH_ABLATE      // the row-shaped first versions (A/B builds only)
__global__ __launch_bounds__(256) void ace_interior_f32_kernel(const AceInteriorParams q) {
    constexpr int RS = 2 * IN_CG + 1;
    __shared__ float gt[19 * RS];
    __shared__ float pa[IN_CG], pd[IN_CG], pn[IN_CG];
    const int HW = q.H * q.W, ppb = (HW + 1023) / 1024;
    const int b = blockIdx.x / ppb, c0 = blockIdx.y * IN_CG;
    const int pix = (blockIdx.x % ppb) * 1024 + threadIdx.x * 4;          // W % 4 == 0: the four pixels share a row
    uchar4 j4 = make_uchar4(255, 255, 255, 255);
    if (pix < HW) j4 = *reinterpret_cast<const uchar4*>(q.u5 + (long long)b * HW + pix);
    const bool i0 = j4.x < 19, i1 = j4.y < 19, i2 = j4.z < 19, i3 = j4.w < 19;
    if (__syncthreads_or(i0 || i1 || i2 || i3) == 0) return;             // no interior pixel in this block
    for (int i = threadIdx.x; i < 19 * 2 * IN_CG; i += 256) {
        const int jj = i / (2 * IN_CG), r = i % (2 * IN_CG), gb = r / IN_CG, c = c0 + r % IN_CG;
        gt[jj * RS + r] = c < q.C ? q.gtab[(((long long)b * 19 + jj) * 2 + gb) * q.C + c] : 0.f;
    }
    if (threadIdx.x < IN_CG) {
        const int c = c0 + threadIdx.x;
        pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
        pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
        pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
    }
    __syncthreads();
    if (!(i0 || i1 || i2 || i3)) return;
    const int y = pix / q.W, x = pix - y * q.W;
    const float* np = q.noise + (long long)b * q.noise_bstride + (long long)x * q.H + y;      // plane layout [W][H]
    const float nz0 = np[0], nz1 = np[q.H], nz2 = np[2 * q.H], nz3 = np[3 * q.H];
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up);
    const float* xp = q.x + ((long long)b * q.C + c0) * xHW + (y >> q.x_up) * xW + (x >> q.x_up);
    float* op = reinterpret_cast<float*>(q.out) + ((long long)b * q.C + c0) * HW + pix;
    const float *g0 = gt + (i0 ? j4.x : 0) * RS, *g1 = gt + (i1 ? j4.y : 0) * RS, *g2 = gt + (i2 ? j4.z : 0) * RS,
                *g3 = gt + (i3 ? j4.w : 0) * RS;
    const int cmax = q.C - c0 < IN_CG ? q.C - c0 : IN_CG;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
    const bool all4 = i0 && i1 && i2 && i3;
#pragma unroll 4
    for (int c = 0; c < cmax; ++c) {
        float4 xv;
        if (q.x_up) {
            const float2 t = *reinterpret_cast<const float2*>(xp + (long long)c * xHW);
            xv = make_float4(t.x, t.x, t.y, t.y);
        } else {
            xv = *reinterpret_cast<const float4*>(xp + (long long)c * xHW);
        }
        const float a = pa[c], n = pn[c], d = pd[c];
        float4 o;
        o.x = (a * xv.x + n * nz0 + d) * (1.f + g0[c]) + g0[IN_CG + c];
        o.y = (a * xv.y + n * nz1 + d) * (1.f + g1[c]) + g1[IN_CG + c];
        o.z = (a * xv.z + n * nz2 + d) * (1.f + g2[c]) + g2[IN_CG + c];
        o.w = (a * xv.w + n * nz3 + d) * (1.f + g3[c]) + g3[IN_CG + c];
        o.x = fmaxf(o.x, slope * o.x); o.y = fmaxf(o.y, slope * o.y);
        o.z = fmaxf(o.z, slope * o.z); o.w = fmaxf(o.w, slope * o.w);
        float* oc = op + (long long)c * HW;
        if (all4) {
            *reinterpret_cast<float4*>(oc) = o;
        } else {
            if (i0) oc[0] = o.x;
            if (i1) oc[1] = o.y;
            if (i2) oc[2] = o.z;
            if (i3) oc[3] = o.w;
        }
    }
}

// One pixel per thread (4-byte accesses, 256 contiguous bytes per wave instruction) -- the default: measured faster than the
// four-pixels-per-thread kernel above (22.4 vs 24.4 ms per 5 steps at B = 16, 512^2) and than writing whole 32-byte sectors
// (variant 2, 24.0 ms), so neither access width nor partial sectors bound this pass.
__global__ __launch_bounds__(256) void ace_interior_f32_scalar_kernel(const AceInteriorParams q) {
    constexpr int RS = 2 * IN_CG + 1;
    __shared__ float gt[19 * RS];
    __shared__ float pa[IN_CG], pd[IN_CG], pn[IN_CG];
    const int HW = q.H * q.W, ppb = (HW + 255) / 256;
    const int b = blockIdx.x / ppb, c0 = blockIdx.y * IN_CG;
    const int pix = (blockIdx.x % ppb) * 256 + threadIdx.x;
    int j = pix < HW ? q.u5[(long long)b * HW + pix] : 255;
    if (__syncthreads_or(j < 19) == 0) return;             // no interior pixel in this block
    if (q.variant == 2) {
        // Full 32-byte sectors: every pixel of an aligned group of 8 that holds an interior pixel is written (a boundary pixel
        // of the group gets a placeholder computed with its neighbour's table row; the boundary conv, which runs after this
        // kernel on the same stream, overwrites it) -- partial-sector writes cost a read-modify-write in ECC memory.
        const unsigned long long m = __ballot(j < 19);
        const int lane = threadIdx.x & 63, l8 = lane & ~7;
        const unsigned grp = (unsigned)((m >> l8) & 0xFFull);
        const int jn = __shfl(j, grp ? l8 + __ffs((int)grp) - 1 : lane, 64);     // first interior lane of the group
        if (j >= 19 && grp != 0u && pix < HW) j = jn;
    }
    for (int i = threadIdx.x; i < 19 * 2 * IN_CG; i += 256) {
        const int jj = i / (2 * IN_CG), r = i % (2 * IN_CG), gb = r / IN_CG, c = c0 + r % IN_CG;
        gt[jj * RS + r] = c < q.C ? q.gtab[(((long long)b * 19 + jj) * 2 + gb) * q.C + c] : 0.f;
    }
    if (threadIdx.x < IN_CG) {
        const int c = c0 + threadIdx.x;
        pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
        pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
        pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
    }
    __syncthreads();
    if (j >= 19) return;
    const int y = pix / q.W, x = pix - y * q.W;
    const float nz = q.noise[(long long)b * q.noise_bstride + (long long)x * q.H + y];
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up);
    const float* xp = q.x + ((long long)b * q.C + c0) * xHW + (y >> q.x_up) * xW + (x >> q.x_up);
    float* op = reinterpret_cast<float*>(q.out) + ((long long)b * q.C + c0) * HW + pix;
    const float* g = gt + j * RS;
    const int cmax = q.C - c0 < IN_CG ? q.C - c0 : IN_CG;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
#pragma unroll 8
    for (int c = 0; c < cmax; ++c) {
        const float xv = xp[(long long)c * xHW];
        const float nrm = pa[c] * xv + pn[c] * nz + pd[c];
        float o = nrm * (1.f + g[c]) + g[IN_CG + c];
        o = fmaxf(o, slope * o);
        op[(long long)c * HW] = o;
    }
}
#endif   // CH_ABLATE

// Blocks of 32 x 8 pixels (the default; tools/interior_bench.hip measures it against the row kernel above): whole 32-byte
// sectors of the [W][H] noise plane instead of 4 bytes out of each of 256 lines, and a block that is mostly interior writes ALL
// its pixels -- no divergence, no byte-masked sectors; the boundary conv runs after this pass on the same stream and
// overwrites the boundary pixels (a filler pixel is computed with table row 0).
__global__ __launch_bounds__(256) void ace_interior_f32_tile_kernel(const AceInteriorParams q) {
    constexpr int RS = 2 * IN_CG + 1;
    __shared__ float gt[19 * RS];
    __shared__ float pa[IN_CG], pd[IN_CG], pn[IN_CG];
    const int HW = q.H * q.W, tpr = (q.W + 31) >> 5, tpc = (q.H + 7) >> 3;
    const int b = blockIdx.x / (tpr * tpc), r = blockIdx.x - b * (tpr * tpc), tyi = r / tpr, c0 = blockIdx.y * IN_CG;
    const int x = (r - tyi * tpr) * 32 + (threadIdx.x & 31), y = tyi * 8 + (threadIdx.x >> 5);
    const bool inimg = x < q.W && y < q.H;
    const int pix = y * q.W + x;
    int j = inimg ? q.u5[(long long)b * HW + pix] : 255;
    if (q.quad_only && inimg) {          // overlap mode: the boundary conv writes every pixel of a boundary quad at the same time
        const uint8_t* uq = q.u5 + (long long)b * HW + (y & ~1) * q.W + (x & ~1);
        if (uq[0] >= 19 || uq[1] >= 19 || uq[q.W] >= 19 || uq[q.W + 1] >= 19) j = 255;
    }
    const bool mine = j < 19;
    const int nmine = __syncthreads_count(mine);
    if (nmine == 0) return;                                  // no interior pixel in this block
    const bool wr = inimg && (mine || nmine >= (q.fill_min > 0 ? q.fill_min : 128));
    if (!mine) j = 0;
    const float nz = wr ? q.noise[(long long)b * q.noise_bstride + (long long)x * q.H + y] : 0.f;
    for (int i = threadIdx.x; i < 19 * 2 * IN_CG; i += 256) {
        const int jj = i / (2 * IN_CG), rr = i % (2 * IN_CG), gb = rr / IN_CG, c = c0 + rr % IN_CG;
        gt[jj * RS + rr] = c < q.C ? q.gtab[(((long long)b * 19 + jj) * 2 + gb) * q.C + c] : 0.f;
    }
    if (threadIdx.x < IN_CG) {
        const int c = c0 + threadIdx.x;
        pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
        pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
        pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
    }
    __syncthreads();
    if (!wr) return;
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up);
    const float* __restrict__ xp = q.x + ((long long)b * q.C + c0) * xHW + (y >> q.x_up) * xW + (x >> q.x_up);
    float* __restrict__ op = reinterpret_cast<float*>(q.out) + ((long long)b * q.C + c0) * HW + pix;
    const float* g = gt + j * RS;
    const int cmax = q.C - c0 < IN_CG ? q.C - c0 : IN_CG;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
#pragma unroll 8
    for (int c = 0; c < cmax; ++c) {
        const float xv = xp[(long long)c * xHW];
        const float nrm = pa[c] * xv + pn[c] * nz + pd[c];
        float o = nrm * (1.f + g[c]) + g[IN_CG + c];
        o = fmaxf(o, slope * o);
        op[(long long)c * HW] = o;
    }
}

#ifndef ACE_T4_AHEAD
#define ACE_T4_AHEAD 0      // x of channel c loaded this many iterations early: 0 = 68 VGPRs, 1 = 80, 2 = 90, 3 = 100: 4.14 / 4.25 / 5.63 / 6.18 ms per step -- the
#endif                      // pass lives on occupancy, not on loads in flight per wave (round 6, same-box A/B; kept as a switch)
#ifndef ACE_T4_UNROLL
#define ACE_T4_UNROLL 1      // channel-loop unroll of the four-pixel interior kernel: 1 = 68 VGPRs (seven blocks per CU), 4 = 97 (five): 4.0 vs 5.3 ms per step (profiles/r06_interior_ab.txt)
#endif
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));      // (a 16-byte store the compiler keeps whole)
// Four pixels of a row per thread, blocks of 128 x 8 pixels (W >= 128): 16-byte stores, 8- / 16-byte x loads, the four noise
// values of a thread share their 32-byte sectors with the seven other rows of the block.
__global__ __launch_bounds__(256, 8) void ace_interior_f32_tile4_kernel(const AceInteriorParams q) {      // (8 blocks per CU: <= 64 VGPR; it is a streaming kernel)
#ifndef ACE_T4_ES_BITS
#define ACE_T4_ES_BITS 6      // 64 slots.  32 (14 KB of LDS instead of 22: eight blocks per CU at 64 VGPRs) was measured: 5.6 vs 4.2 ms per step on the
#endif                        // benchmark labels -- a 128 x 8 block of the 128-pixel level holds up to ~60 codes, so it takes a second round; 16: 9.1 ms
    constexpr int RS = 2 * IN_CG + 1, ES = 1 << ACE_T4_ES_BITS;      // ES: slots of the block's table of straight-edge rows
    __shared__ float gt[19 * RS];
    __shared__ float et[ES * RS];
    __shared__ int ekey[ES], olist[ES], nocc;
    __shared__ float pa[IN_CG], pd[IN_CG], pn[IN_CG];
    const int HW = q.H * q.W, tpr = (q.W + 127) >> 7, tpc = (q.H + 7) >> 3;
    const int b = blockIdx.x / (tpr * tpc), r = blockIdx.x - b * (tpr * tpc), tyi = r / tpr, c0 = blockIdx.y * IN_CG;
    const int x = (r - tyi * tpr) * 128 + (threadIdx.x & 31) * 4, y = tyi * 8 + (threadIdx.x >> 5);
    const bool inimg = x < q.W && y < q.H;                   // W % 4 == 0: the four pixels are inside together
    const int pix = y * q.W + x;
    uchar4 j4 = make_uchar4(255, 255, 255, 255);
    if (inimg) j4 = *reinterpret_cast<const uchar4*>(q.u5 + (long long)b * HW + pix);
    bool i0 = j4.x < 19, i1 = j4.y < 19, i2 = j4.z < 19, i3 = j4.w < 19;
    // straight-edge pixels (u5 == 253, ace_sparse.h): modulated here as well, with the table row of their code (+ three style sums)
    const bool edges = q.e16 != nullptr && !q.quad_only;
    const bool e0 = edges && j4.x == ACE_EDGE, e1 = edges && j4.y == ACE_EDGE, e2 = edges && j4.z == ACE_EDGE, e3 = edges && j4.w == ACE_EDGE;
    if (q.quad_only && inimg) {          // overlap mode: the boundary conv writes every pixel of a boundary quad at the same time
        const uchar4 o4 = *reinterpret_cast<const uchar4*>(q.u5 + (long long)b * HW + (y ^ 1) * q.W + x);      // the other row of the two quads
        const bool q0 = i0 && i1 && o4.x < 19 && o4.y < 19, q1 = i2 && i3 && o4.z < 19 && o4.w < 19;
        i0 = i1 = q0;
        i2 = i3 = q1;
    }
    const bool m0 = i0 || e0, m1 = i1 || e1, m2 = i2 || e2, m3 = i3 || e3;      // pixels this pass owns
    const int nmine = __syncthreads_count(m0) + __syncthreads_count(m1) + __syncthreads_count(m2) + __syncthreads_count(m3);
    if (nmine == 0) return;                                  // no pixel of this pass in this block
    // What is written.  A partially written 128-byte line costs more than a whole one (masked stores: 538 vs 423 us at 77 % interior
    // pixels, profiles/r05_interior_bench.txt), so boundary pixels next to owned ones are written too -- the boundary conv, launched
    // after this pass, overwrites them.  fill_min = 0 (default since round 6): per LINE -- the eight threads of a 128-byte line write
    // it whole if it holds an owned pixel and skip it (loads and stores) if it holds none: the all-boundary rows along the region
    // borders, 12 / 25 / 50 % of the lines of the 512 / 256 / 128-pixel levels on the benchmark labels, are no longer written twice.
    // fill_min > 0 (rounds 3-5): per BLOCK of 128 x 8 pixels -- all of it if it has at least 4 fill_min owned pixels, else masked.
    bool fill;
    if (q.fill_min == 0) {
        const unsigned long long bal = __ballot(m0 || m1 || m2 || m3);
        fill = ((bal >> (threadIdx.x & 56)) & 0xFFull) != 0;
    } else {
        fill = nmine >= 4 * q.fill_min;
    }
    const bool any = inimg && (fill || m0 || m1 || m2 || m3);
    float nz0 = 0.f, nz1 = 0.f, nz2 = 0.f, nz3 = 0.f;
    if (any) {
        const float* np = q.noise + (long long)b * q.noise_bstride + (long long)x * q.H + y;      // plane layout [W][H]
        nz0 = np[0]; nz1 = np[q.H]; nz2 = np[2 * q.H]; nz3 = np[3 * q.H];
    }
    for (int i = threadIdx.x; i < 19 * 2 * IN_CG; i += 256) {
        const int jj = i / (2 * IN_CG), rr = i % (2 * IN_CG), gb = rr / IN_CG, c = c0 + rr % IN_CG;
        gt[jj * RS + rr] = c < q.C ? q.gtab[(((long long)b * 19 + jj) * 2 + gb) * q.C + c] : 0.f;
    }
    if (threadIdx.x < IN_CG) {
        const int c = c0 + threadIdx.x;
        pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
        pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
        pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
    }
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up);
    const float* __restrict__ xp = q.x + ((long long)b * q.C + c0) * xHW + (y >> q.x_up) * xW + (x >> q.x_up);
    float* __restrict__ op = reinterpret_cast<float*>(q.out) + ((long long)b * q.C + c0) * HW + pix;
    const int cmax = q.C - c0 < IN_CG ? q.C - c0 : IN_CG;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
    // Straight-edge pixels: the block's distinct codes go into a 64-slot table (open addressing on the code, LDS compare-and-swap); the
    // rows -- E[code] + the three style sums -- are built once per block, cooperatively, and read like the interior rows.  Codes that
    // find no slot (more than 64 distinct ones in 128 x 8 pixels: not seen on any label map of the tests) wait for another ROUND of the
    // same code with a fresh table; a later round stores its pixels one by one.  pend: the edge pixels without a row yet.
    unsigned short kk[4] = {0, 0, 0, 0};
    bool pend[4] = {e0, e1, e2, e3};
    if (e0 || e1 || e2 || e3) {
        const ushort4 k4 = *reinterpret_cast<const ushort4*>(q.e16 + (long long)b * HW + pix);
        kk[0] = k4.x; kk[1] = k4.y; kk[2] = k4.z; kk[3] = k4.w;
    }
    for (bool first = true;; first = false) {
        int slot[4] = {-1, -1, -1, -1};
        if (edges) {
            if (threadIdx.x < ES) ekey[threadIdx.x] = -1;
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (!pend[s]) continue;
                const int code = kk[s];
                unsigned h = ((unsigned)code * 2654435761u) >> (32 - ACE_T4_ES_BITS);
                for (int probe = 0; probe < ES; ++probe) {
                    const int old = atomicCAS(&ekey[h], -1, code);
                    if (old == -1 || old == code) {
                        slot[s] = (int)h;
                        pend[s] = false;
                        break;
                    }
                    h = (h + 1) & (ES - 1);
                }
            }
            __syncthreads();                                 // (the slots are final)
            if (threadIdx.x < ES) {                          // the occupied slots, compacted (most blocks hold 0 ... 20 codes)
                const bool occ = ekey[threadIdx.x] >= 0;
                const unsigned long long om = __ballot(occ);
                if (occ) olist[__popcll(om & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
                if (threadIdx.x == 0) nocc = __popcll(om);
            }
            __syncthreads();
            const int ne = nocc * 2 * IN_CG;
            for (int i = threadIdx.x; i < ne; i += 256) {
                const int sl = olist[i / (2 * IN_CG)], rr = i % (2 * IN_CG), gb = rr / IN_CG, c = c0 + rr % IN_CG;
                const int code = ekey[sl];
                float v = 0.f;
                if (c < q.C) {
                    v = q.etab[((long long)code * 2 + gb) * q.C + c];
                    if (q.p6) {
                        const int sn = (code & 3) + 1, Bl = (code >> 2) % 19, A = ((code >> 2) / 19) % 19, o = (code >> 2) / 361;
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            const int l = (1 + t < sn) ? A : Bl;        // line label 1 + t of A^s B^(5-s)
                            v += q.p6[((((long long)b * 19 + l) * 6 + o * 3 + t) * 2 + gb) * q.C + c];
                        }
                    }
                }
                et[sl * RS + rr] = v;
            }
        }
        __syncthreads();
        // this round's pixels: the first round takes the interior pixels, the edge pixels that found a slot and (fill) the line's others
        const bool w0 = first ? (i0 || slot[0] >= 0) : slot[0] >= 0, w1 = first ? (i1 || slot[1] >= 0) : slot[1] >= 0,
                   w2 = first ? (i2 || slot[2] >= 0) : slot[2] >= 0, w3 = first ? (i3 || slot[3] >= 0) : slot[3] >= 0;
        if (first ? any : (w0 || w1 || w2 || w3)) {
            // a pixel's gamma | beta row: the (sample, label) row of an interior pixel, the block-table row of a straight-edge pixel
            const float *g0 = slot[0] >= 0 ? et + slot[0] * RS : gt + (i0 ? j4.x : 0) * RS, *g1 = slot[1] >= 0 ? et + slot[1] * RS : gt + (i1 ? j4.y : 0) * RS,
                        *g2 = slot[2] >= 0 ? et + slot[2] * RS : gt + (i2 ? j4.z : 0) * RS, *g3 = slot[3] >= 0 ? et + slot[3] * RS : gt + (i3 ? j4.w : 0) * RS;
            // x of channel c (optionally loaded ACE_T4_AHEAD iterations before its use: measured slower, see the macro)
            auto loadx = [&](int c) {
                float4 xv;
                if (q.x_up) {
                    const float2 t = *reinterpret_cast<const float2*>(xp + (long long)c * xHW);
                    xv = make_float4(t.x, t.x, t.y, t.y);
                } else {
                    xv = *reinterpret_cast<const float4*>(xp + (long long)c * xHW);
                }
                return xv;
            };
            auto channel = [&](int c, const float4 xv) {
                const float a = pa[c], n = pn[c], d = pd[c];
                float4 o;
                o.x = (a * xv.x + n * nz0 + d) * (1.f + g0[c]) + g0[IN_CG + c];
                o.y = (a * xv.y + n * nz1 + d) * (1.f + g1[c]) + g1[IN_CG + c];
                o.z = (a * xv.z + n * nz2 + d) * (1.f + g2[c]) + g2[IN_CG + c];
                o.w = (a * xv.w + n * nz3 + d) * (1.f + g3[c]) + g3[IN_CG + c];
                o.x = fmaxf(o.x, slope * o.x); o.y = fmaxf(o.y, slope * o.y);
                o.z = fmaxf(o.z, slope * o.z); o.w = fmaxf(o.w, slope * o.w);
                return o;
            };
            // Two loops, not one loop with the choice inside: with `if (all4) 16-byte store else four masked stores` in one body hipcc
            // if-converts both arms into four predicated 4-byte stores -- the shipped kernel of rounds 4-5 never issued a
            // global_store_dwordx4 (found in round 6 when an unrelated branch in the body changed the code: 3.37 -> 2.75 ms per step).
            constexpr int QN = ACE_T4_AHEAD > 0 ? ACE_T4_AHEAD : 1;
            float4 xq[QN];
            if constexpr (ACE_T4_AHEAD > 0) {
#pragma unroll
                for (int k = 0; k < QN; ++k) xq[k] = loadx(k < cmax ? k : cmax - 1);
            }
            auto next_x = [&](int c) {                  // x of channel c; the queue moves on to channel c + ACE_T4_AHEAD
                if constexpr (ACE_T4_AHEAD == 0) return loadx(c);
                const float4 xv = xq[0];
#pragma unroll
                for (int k = 0; k + 1 < QN; ++k) xq[k] = xq[k + 1];
                xq[QN - 1] = loadx(c + QN < cmax ? c + QN : cmax - 1);
                return xv;
            };
            if (first && (fill || (w0 && w1 && w2 && w3))) {
#pragma unroll ACE_T4_UNROLL
                for (int c = 0; c < cmax; ++c) {
                    const float4 o = channel(c, next_x(c));
                    *reinterpret_cast<nt_f32x4*>(op + (long long)c * HW) = (nt_f32x4){o.x, o.y, o.z, o.w};
                }
            } else {
#pragma unroll ACE_T4_UNROLL
                for (int c = 0; c < cmax; ++c) {
                    const float4 o = channel(c, next_x(c));
                    float* oc = op + (long long)c * HW;
                    if (w0) oc[0] = o.x;
                    if (w1) oc[1] = o.y;
                    if (w2) oc[2] = o.z;
                    if (w3) oc[3] = o.w;
                }
            }
        }
        if (!edges) break;
        if (!__syncthreads_or(pend[0] || pend[1] || pend[2] || pend[3])) break;      // (block-uniform; also orders this round's table reads before the next clear)
    }
}



hipError_t ace_interior_f32(const AceInteriorParams& q, hipStream_t s) {
    if (q.act > ACT_RELU) return hipErrorInvalidValue;
    if (q.W % 4 != 0) return hipErrorInvalidValue;
    const int HW = q.H * q.W;
    if (q.impl == 2 && q.W >= 128) {
        dim3 gridt((unsigned)(q.B * ((q.W + 127) / 128) * ((q.H + 7) / 8)), (unsigned)((q.C + IN_CG - 1) / IN_CG));
        hipLaunchKernelGGL(ace_interior_f32_tile4_kernel, gridt, dim3(256), 0, s, q);
        return hipGetLastError();
    }
    if (q.impl == 0 || q.impl == 2) {
        dim3 gridt((unsigned)(q.B * ((q.W + 31) / 32) * ((q.H + 7) / 8)), (unsigned)((q.C + IN_CG - 1) / IN_CG));
        hipLaunchKernelGGL(ace_interior_f32_tile_kernel, gridt, dim3(256), 0, s, q);
        return hipGetLastError();
    }
#ifndef CH_ABLATE
    (void)HW;
    return hipErrorInvalidValue;
#else
    if (q.variant != 1) {       // default (0) and the full-sector experiment (2): one pixel per thread
        dim3 grid1((unsigned)(q.B * ((HW + 255) / 256)), (unsigned)((q.C + IN_CG - 1) / IN_CG));
        hipLaunchKernelGGL(ace_interior_f32_scalar_kernel, grid1, dim3(256), 0, s, q);
        return hipGetLastError();
    }
    dim3 grid((unsigned)(q.B * ((HW + 1023) / 1024)), (unsigned)((q.C + IN_CG - 1) / IN_CG));
    hipLaunchKernelGGL(ace_interior_f32_kernel, grid, dim3(256), 0, s, q);
    return hipGetLastError();
#endif
}

// ---- interior pass, f16x3 path (tile-skip mode): C4 in, SH16 out --------------------------------------------------------
// One thread = one pixel x 8 channels (one SH16 unit pair); one block = 256 consecutive pixels x IS_GPB channel groups.
// Only pixels of tiles of 32 x 16 without a boundary pixel are written (the conv kernel writes every pixel of the others).
//      o = act((bn_a x + nv nz + bn_d) (1 + gamma) + beta) * out_scale [* extra];  hi = f16(o), lo = f16(o - hi)
constexpr int IS_GPB = 4;
typedef _Float16 is_h8 __attribute__((ext_vector_type(8)));
#ifdef CH_ABLATE      // the row-shaped first version (A/B builds only)
__global__ __launch_bounds__(256) void ace_interior_sh16_kernel(const AceInteriorParams q) {
    constexpr int CB = IS_GPB * 8, RS = 2 * CB + 4;          // row pitch: 16-byte aligned, labels spread over the banks
    __shared__ __attribute__((aligned(16))) float gt[19 * RS];
    __shared__ __attribute__((aligned(16))) float pa[CB], pd[CB], pn[CB];
    sh16_mode_on();
    float extra = 1.f;
    if (q.pass == 1) {
        extra = sh16_dyn_extra(*q.out_amax);
        if (extra == 1.f) return;                             // nothing to repair (the normal case)
    }
    const int HW = q.H * q.W, ppb = (HW + 255) / 256, nblk = q.B * ppb;
    const int tiles_x = (q.W + 31) >> 5, tiles_y = (q.H + 15) >> 4;
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up), Go = (q.C + 7) >> 3;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
    const float osc = q.out_scale * extra;
    float amax = 0.f;
    // persistent blocks, grid-stride over blocks of 256 consecutive pixels: on label maps without a boundary-free tile (every
    // block leaves after one byte per thread) the launch costs a few microseconds instead of one dispatch per pixel block
    for (int pbk = blockIdx.x; pbk < nblk; pbk += gridDim.x) {
        const int b = pbk / ppb, pix = (pbk - b * ppb) * 256 + threadIdx.x;
        const int y = pix / q.W, x = pix - y * q.W;
        bool mine = false;
        int j = 255;
        if (pix < HW) {
            // variant 0: the pixels of boundary-free tiles (tile-skip mode); 1: every interior pixel (compacting conv kernel)
            mine = q.variant == 1 || q.cnt[(b * tiles_y + (y >> 4)) * tiles_x + (x >> 5)] == 0;
            if (mine) j = q.u5[(long long)b * HW + pix];
            mine = mine && j < 19;
        }
        if (__syncthreads_or(mine) == 0) continue;
        const float nz = mine ? q.noise[(long long)b * q.noise_bstride + (long long)x * q.H + y] : 0.f;
        const long long xpix = (long long)(y >> q.x_up) * xW + (x >> q.x_up);
        const float4* xp = reinterpret_cast<const float4*>(q.x) + (long long)b * (q.C >> 2) * xHW + xpix;
        uint4* op = reinterpret_cast<uint4*>(q.out) + (long long)b * Go * 2 * HW + pix;
        for (int c0 = 0; c0 < q.C; c0 += CB) {
            __syncthreads();                                  // the previous slice is no longer read
            for (int i = threadIdx.x; i < 19 * 2 * CB; i += 256) {
                const int jj = i / (2 * CB), r = i % (2 * CB), gb = r / CB, c = c0 + r % CB;
                gt[jj * RS + r] = c < q.C ? q.gtab[(((long long)b * 19 + jj) * 2 + gb) * q.C + c] : 0.f;
            }
            if (threadIdx.x < CB) {
                const int c = c0 + threadIdx.x;
                pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
                pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
                pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
            }
            __syncthreads();
            if (!mine) continue;
            const float* g = gt + j * RS;
#pragma unroll
            for (int gq = 0; gq < IS_GPB; ++gq) {
                const int c = c0 + gq * 8;
                if (c >= q.C) break;
                float xv[8];
                {
                    const float4 a = xp[(long long)(c >> 2) * xHW];
                    xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w;
                    if (c + 4 < q.C) {
                        const float4 d = xp[(long long)((c >> 2) + 1) * xHW];
                        xv[4] = d.x; xv[5] = d.y; xv[6] = d.z; xv[7] = d.w;
                    } else {
                        xv[4] = xv[5] = xv[6] = xv[7] = 0.f;
                    }
                }
                is_h8 vh, vl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cc = gq * 8 + e;
                    const float nrm = pa[cc] * xv[e] + pn[cc] * nz + pd[cc];
                    float o = nrm * (1.f + g[cc]) + g[CB + cc];
                    o = fmaxf(o, slope * o) * osc;
                    if (c + e >= q.C) o = 0.f;                 // padding channels of the last group hold zeros
                    amax = fmaxf(amax, fabsf(o));
                    if (q.bf16) {
                        const __bf16 t = (__bf16)o;
                        vh[e] = __builtin_bit_cast(_Float16, t);
                        vl[e] = (_Float16)0.f;
                    } else {
                        const _Float16 h = (_Float16)o;
                        vh[e] = h;
                        vl[e] = (_Float16)(o - (float)h);
                    }
                }
                const long long u = (long long)(c >> 3) * 2 * HW;
                op[u] = __builtin_bit_cast(uint4, vh);
                if (!q.single) op[u + HW] = __builtin_bit_cast(uint4, vl);
            }
        }
    }
    // pass 0: the tensor's maximum at the first-pass scale (uniform: every thread of every block gets here)
    if (q.pass == 0 && q.out_amax) sh16_block_slot_max(q.out_amax, amax);
}
#endif   // CH_ABLATE

// Same arithmetic, blocks of 32 x 8 pixels (tools/interior_bench.hip measures both):
//  * the noise plane is stored [W][H]: a block of 256 consecutive pixels of a row touched 256 different 64-byte lines of it for
//    4 bytes each (and the 15 other rows of a line sit in blocks that run on other XCDs, behind other L2s); 8 rows of 32 pixels
//    use whole 32-byte sectors;
//  * the slab's x loads are issued BEFORE the table staging and its barriers (they do not depend on the table), so a slab
//    exposes one memory latency instead of two; the table is staged with 16-byte loads;
//  * pixel-level mode: a block that is mostly interior writes ALL its pixels (no divergence, no byte-masked partial sectors);
//    the boundary conv runs after this pass on the same stream and overwrites the boundary pixels.  Those pixels never enter
//    the recorded maximum.
__global__ __launch_bounds__(256) void ace_interior_sh16_tile_kernel(const AceInteriorParams q) {
    constexpr int CB = IS_GPB * 8, RS = 2 * CB + 4, ES = 64;
    __shared__ __attribute__((aligned(16))) float gt[19 * RS];
    __shared__ __attribute__((aligned(16))) float et[ES * RS];      // rows of the tile's straight-edge codes (round 6, ace_sparse.h), per channel slab
    __shared__ int ekey[ES], olist[ES], nocc;
    __shared__ __attribute__((aligned(16))) float pa[CB], pd[CB], pn[CB];
    sh16_mode_on();
    float extra = 1.f;
    if (q.pass == 1) {
        extra = sh16_dyn_extra(*q.out_amax);
        if (extra == 1.f) return;                             // nothing to repair (the normal case)
    }
    const int HW = q.H * q.W;
    const int tpr = (q.W + 31) >> 5, tpc = (q.H + 7) >> 3, ntile = q.B * tpr * tpc;
    const int tiles_x = tpr, tiles_y = (q.H + 15) >> 4;
    const int xW = q.W >> q.x_up, xHW = xW * (q.H >> q.x_up), Go = (q.C + 7) >> 3;
    const float slope = q.act == ACT_NONE ? 1.f : (q.act == ACT_LRELU ? 0.2f : 0.f);
    const float osc = q.out_scale * extra;
    const int fill_min = q.variant == 1 ? (q.fill_min > 0 ? q.fill_min : 128) : 257;
    const bool edges = q.e16 != nullptr && q.etab != nullptr;
    float amax = 0.f;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int b = t / (tpr * tpc), r = t - b * (tpr * tpc), tyi = r / tpr;
        const int x = (r - tyi * tpr) * 32 + (threadIdx.x & 31), y = tyi * 8 + (threadIdx.x >> 5);
        const bool inimg = x < q.W && y < q.H;
        const int pix = y * q.W + x;
        bool mine = false, edge = false;
        int j = 255, code = 0;
        if (inimg) {
            mine = q.variant == 1 || q.cnt[(b * tiles_y + (y >> 4)) * tiles_x + (x >> 5)] == 0;
            if (mine) j = q.u5[(long long)b * HW + pix];
            edge = edges && mine && j == ACE_EDGE;            // a straight-edge pixel: its row comes from the tile's table of codes
            if (edge) code = q.e16[(long long)b * HW + pix];
            mine = mine && j < 19;
        }
        const bool own = mine || edge;
        const int nmine = __syncthreads_count(own);
        if (nmine == 0) continue;
        const bool wr0 = inimg && (own || nmine >= fill_min);
        if (!mine) j = 0;
        const float nz = wr0 ? q.noise[(long long)b * q.noise_bstride + (long long)x * q.H + y] : 0.f;
        const long long xpix = (long long)(y >> q.x_up) * xW + (x >> q.x_up);
        const float4* __restrict__ xp = reinterpret_cast<const float4*>(q.x) + (long long)b * (q.C >> 2) * xHW + (inimg ? xpix : 0);
        uint4* __restrict__ op = reinterpret_cast<uint4*>(q.out) + (long long)b * Go * 2 * HW + pix;
        // rounds: the codes that find no slot in the 64-entry table of a round are served by the next one (not seen on the tests' label maps)
        bool pend = edge;
        for (bool first = true;; first = false) {
            int slot = -1;
            if (edges) {
                __syncthreads();                              // (the previous tile's / round's table is no longer read)
                if (threadIdx.x < ES) ekey[threadIdx.x] = -1;
                __syncthreads();
                if (pend) {
                    unsigned h = ((unsigned)code * 2654435761u) >> 26;
                    for (int probe = 0; probe < ES; ++probe) {
                        const int old = atomicCAS(&ekey[h], -1, code);
                        if (old == -1 || old == code) {
                            slot = (int)h;
                            pend = false;
                            break;
                        }
                        h = (h + 1) & (ES - 1);
                    }
                }
                __syncthreads();
                if (threadIdx.x < ES) {
                    const bool occ = ekey[threadIdx.x] >= 0;
                    const unsigned long long om = __ballot(occ);
                    if (occ) olist[__popcll(om & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
                    if (threadIdx.x == 0) nocc = __popcll(om);
                }
            }
            const bool wr = first ? wr0 : slot >= 0;          // later rounds: only the pixels that just got their row
            for (int c0 = 0; c0 < q.C; c0 += CB) {
                float4 xa[IS_GPB][2];
#pragma unroll
                for (int gq = 0; gq < IS_GPB; ++gq) {
                    const int c = c0 + gq * 8;
                    xa[gq][0] = xa[gq][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (wr && c < q.C) {
                        xa[gq][0] = xp[(long long)(c >> 2) * xHW];
                        if (c + 4 < q.C) xa[gq][1] = xp[(long long)((c >> 2) + 1) * xHW];
                    }
                }
                __syncthreads();                              // the previous slab's tables are no longer read
                for (int i = threadIdx.x; i < 19 * 2 * (CB / 4); i += 256) {
                    const int jj = i / (2 * (CB / 4)), r4 = i % (2 * (CB / 4)), gb = r4 / (CB / 4), c = c0 + (r4 % (CB / 4)) * 4;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c < q.C) v = *reinterpret_cast<const float4*>(q.gtab + (((long long)b * 19 + jj) * 2 + gb) * q.C + c);
                    *reinterpret_cast<float4*>(gt + jj * RS + gb * CB + (r4 % (CB / 4)) * 4) = v;
                }
                if (edges) {                                  // rows of the occupied slots: E[code] + the three style sums (C % 4 == 0)
                    const int ne = nocc * 2 * (CB / 4);
                    for (int i = threadIdx.x; i < ne; i += 256) {
                        const int sl = olist[i / (2 * (CB / 4))], r4 = i % (2 * (CB / 4)), gb = r4 / (CB / 4), c = c0 + (r4 % (CB / 4)) * 4;
                        const int kc = ekey[sl];
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (c < q.C) {
                            v = *reinterpret_cast<const float4*>(q.etab + ((long long)kc * 2 + gb) * q.C + c);
                            if (q.p6) {
                                const int sn = (kc & 3) + 1, Bl = (kc >> 2) % 19, A = ((kc >> 2) / 19) % 19, o = (kc >> 2) / 361;
#pragma unroll
                                for (int tt = 0; tt < 3; ++tt) {
                                    const int l = (1 + tt < sn) ? A : Bl;
                                    const float4 u = *reinterpret_cast<const float4*>(q.p6 + ((((long long)b * 19 + l) * 6 + o * 3 + tt) * 2 + gb) * q.C + c);
                                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                                }
                            }
                        }
                        *reinterpret_cast<float4*>(et + sl * RS + gb * CB + (r4 % (CB / 4)) * 4) = v;
                    }
                }
                if (threadIdx.x < CB) {
                    const int c = c0 + threadIdx.x;
                    pa[threadIdx.x] = c < q.C ? q.bn_a[c] : 0.f;
                    pd[threadIdx.x] = c < q.C ? q.bn_d[c] : 0.f;
                    pn[threadIdx.x] = c < q.C ? q.nv[c] : 0.f;
                }
                __syncthreads();
                if (!wr) continue;
                const float* g = slot >= 0 ? et + slot * RS : gt + j * RS;
#pragma unroll
                for (int gq = 0; gq < IS_GPB; ++gq) {
                    const int c = c0 + gq * 8;
                    if (c >= q.C) break;
                    const float xv[8] = {xa[gq][0].x, xa[gq][0].y, xa[gq][0].z, xa[gq][0].w, xa[gq][1].x, xa[gq][1].y, xa[gq][1].z, xa[gq][1].w};
                    is_h8 vh, vl;
                    float am = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int cc = gq * 8 + e;
                        const float nrm = pa[cc] * xv[e] + pn[cc] * nz + pd[cc];
                        float o = nrm * (1.f + g[cc]) + g[CB + cc];
                        o = fmaxf(o, slope * o) * osc;
                        if (c + e >= q.C) o = 0.f;                 // padding channels of the last group hold zeros
                        am = fmaxf(am, fabsf(o));
                        if (q.bf16) {
                            const __bf16 tb = (__bf16)o;
                            vh[e] = __builtin_bit_cast(_Float16, tb);
                            vl[e] = (_Float16)0.f;
                        } else {
                            const _Float16 h = (_Float16)o;
                            vh[e] = h;
                            vl[e] = (_Float16)(o - (float)h);
                        }
                    }
                    if (mine || slot >= 0) amax = fmaxf(amax, am);  // filler pixels are overwritten: they do not set the scale
                    const long long u = (long long)(c >> 3) * 2 * HW;
                    op[u] = __builtin_bit_cast(uint4, vh);
                    if (!q.single) op[u + HW] = __builtin_bit_cast(uint4, vl);
                }
            }
            if (!edges) break;
            if (!__syncthreads_or(pend)) break;
        }
    }
    if (q.pass == 0 && q.out_amax) sh16_block_slot_max(q.out_amax, amax);
}

hipError_t ace_interior_sh16(const AceInteriorParams& q, hipStream_t s) {
    if (q.act > ACT_RELU || !q.cnt || (q.C & 3)) return hipErrorInvalidValue;
    const int HW = q.H * q.W;
    if (q.impl == 0) {
        const int ntile = q.B * ((q.W + 31) / 32) * ((q.H + 7) / 8);
        hipLaunchKernelGGL(ace_interior_sh16_tile_kernel, dim3((unsigned)(ntile < 2048 ? ntile : 2048)), dim3(256), 0, s, q);
        return hipGetLastError();
    }
#ifdef CH_ABLATE
    const int nblk = q.B * ((HW + 255) / 256);
    hipLaunchKernelGGL(ace_interior_sh16_kernel, dim3((unsigned)(nblk < 2048 ? nblk : 2048)), dim3(256), 0, s, q);
    return hipGetLastError();
#else
    (void)HW;
    return hipErrorInvalidValue;
#endif
}

}  // namespace chk
