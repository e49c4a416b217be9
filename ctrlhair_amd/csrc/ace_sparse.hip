// ace_sparse.hip -- label classification, work lists, per-label gamma/beta tables and the elementwise interior pass of the
// exact SPADE-interior reduction (ace_sparse.h; reference: normalization.py:108-189,249-257).
#include "ace_sparse.h"

#include "conv_mfma.h"
#include "sh16.h"

namespace chk {

// ---- classification: one block = one tile of 32 x TH pixels of one sample (TH * 32 threads) -------------------------------
// A pixel is INTERIOR iff its label is < 19 and all 25 labels of its 5x5 neighbourhood exist (inside the image) and equal it.
template <int TH>
__global__ __launch_bounds__(32 * TH) void ace_classify_kernel(const uint8_t* __restrict__ lab, uint8_t* __restrict__ u5,
                                                             uint16_t* __restrict__ list, int* __restrict__ cnt, int H, int W,
                                                             int tiles_x, int tiles_y) {
    constexpr int NT = 32 * TH, PW = 36, PH = TH + 4, NW = NT / 64;
    __shared__ uint8_t patch[PH * PW];
    __shared__ int wcnt[NW];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * 32;
    const uint8_t* lb = lab + (long long)b * H * W;
    for (int i = tid; i < PH * PW; i += NT) {
        const int y = y0 - 2 + i / PW, x = x0 - 2 + i % PW;
        patch[i] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? lb[(long long)y * W + x] : (uint8_t)255;
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31, y = y0 + ty, x = x0 + tx;
    const bool inside = y < H && x < W;
    const uint8_t c = patch[(ty + 2) * PW + tx + 2];
    bool uni = c < 19;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) uni = uni && patch[(ty + dy) * PW + tx + dx] == c;
    if (inside) u5[(long long)b * H * W + (long long)y * W + x] = uni ? c : (uint8_t)255;
    const bool bnd = inside && !uni;
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

hipError_t ace_classify(const uint8_t* lab, uint8_t* u5, uint16_t* list, int* cnt, int B, int H, int W, int TH, hipStream_t s) {
    const int tx = (W + 31) / 32, ty = (H + TH - 1) / TH;
    if (TH == 8) hipLaunchKernelGGL(ace_classify_kernel<8>, dim3(B * tx * ty), dim3(256), 0, s, lab, u5, list, cnt, H, W, tx, ty);
    else if (TH == 16) hipLaunchKernelGGL(ace_classify_kernel<16>, dim3(B * tx * ty), dim3(512), 0, s, lab, u5, list, cnt, H, W, tx, ty);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- work list: exclusive scan of the block tasks per tile, one block of 1024 threads ------------------------------------
__global__ __launch_bounds__(1024) void ace_worklist_kernel(const int* __restrict__ cnt, int ntiles, int mtiles,
                                                            unsigned* __restrict__ work, int* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    __shared__ int stat[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    if (tid < 3) stat[tid] = 0;
    __syncthreads();
    int s_px = 0, s_sub = 0, s_ws = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int tile = t0 + tid;
        int nbt = 0;
        if (tile < ntiles) {
            const int c = cnt[tile], NS = (c + 31) >> 5;
            int ng, per;
            sparse_groups(NS, mtiles, ng, per);
            nbt = (ng * mtiles + 3) >> 2;
            s_px += c;
            s_sub += NS;
            s_ws += NS * mtiles;
        }
        int v = nbt;                                   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(v, off, 64);
            if (lane >= off) v += u;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int excl = base + v - nbt;
        for (int i = 0; i < nbt; ++i) work[excl + i] = (unsigned)tile | ((unsigned)i << 20);
        __syncthreads();
        if (tid == 1023) carry = base + v;
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
    }
}

hipError_t ace_worklist(const int* cnt, int ntiles, int mtiles, unsigned* work, int* total, hipStream_t s) {
    if (ntiles >= (1 << 20)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ace_worklist_kernel, dim3(1), dim3(1024), 0, s, cnt, ntiles, mtiles, work, total);
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
        for (int t = 0; t < 9; ++t) sacc += lut[(long long)((t * 2 + gb) * C + c) * lut_rs + col];
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

// ---- interior pass, exact-f32 path (NCHW in, NCHW out): HBM-bound streaming ----------------------------------------------
// One block = 256 consecutive pixels of one sample x CG channels; the sample's table slice [19][2][CG] sits in LDS (row
// pitch padded by one float: lanes that hold different labels hit different banks).
//      out = act((bn_a x + nv nz + bn_d) (1 + gamma) + beta)          (normalization.py:111-112,182; architecture.py:95)
constexpr int IN_CG = 32;
__global__ __launch_bounds__(256) void ace_interior_f32_kernel(const AceInteriorParams q) {
    constexpr int RS = 2 * IN_CG + 1;
    __shared__ float gt[19 * RS];
    __shared__ float pa[IN_CG], pd[IN_CG], pn[IN_CG];
    const int HW = q.H * q.W, ppb = (HW + 255) / 256;
    const int b = blockIdx.x / ppb, c0 = blockIdx.y * IN_CG;
    const int pix = (blockIdx.x % ppb) * 256 + threadIdx.x;
    const int j = pix < HW ? q.u5[(long long)b * HW + pix] : 255;
    if (__syncthreads_or(j < 19) == 0) return;             // no interior pixel in this block
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

hipError_t ace_interior_f32(const AceInteriorParams& q, hipStream_t s) {
    if (q.act > ACT_RELU) return hipErrorInvalidValue;
    const int HW = q.H * q.W;
    dim3 grid((unsigned)(q.B * ((HW + 255) / 256)), (unsigned)((q.C + IN_CG - 1) / IN_CG));
    hipLaunchKernelGGL(ace_interior_f32_kernel, grid, dim3(256), 0, s, q);
    return hipGetLastError();
}

}  // namespace chk
