// misc_kernels.hip -- HBM/latency-bound kernels of the Zencoder, shape branch, BiSeNet and colour MLPs:
// wave64 shuffle reductions for the normalisation statistics, pooling, up-sampling/argmax epilogues, GEMV.
#include "kernels.h"
#include "sh16.h"

namespace chk {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// block (256 threads) sum, result broadcast to all threads; `red` = 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.2f * v;
        case 2: return v > 0.f ? v : 0.f;
        case 3: return tanhf(v);
        case 4: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// nn.InstanceNorm2d(affine=False, eps=1e-5) + activation, in place (architecture.py:158-172).  One block per
// (b,c) plane, two-pass statistics (mean, then biased variance), third pass writes.
// sh16 != nullptr: the normalised plane is written (instead of in place) as f16 hi/lo in the SH16 layout
// [B][C/8][2][HW][8] of conv_sh16.h (plane index = b*C + c), feeding the f16x3 conv directly.
__global__ __launch_bounds__(256) void instnorm_act_kernel(float* __restrict__ x, int HW, float eps, int act,
                                                           _Float16* __restrict__ sh16, int C, float scale) {
    sh16_mode_on();
    __shared__ float red[4];
    float* p = x + (long long)blockIdx.x * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    const float mean = block_sum(s, red) / HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float d = p[i] - mean;
        q += d * d;
    }
    const float var = block_sum(q, red) / HW;
    const float rstd = 1.f / sqrtf(var + eps);
    if (sh16) {
        const int b = blockIdx.x / C, c = blockIdx.x % C;
        _Float16* oh = sh16 + ((((long long)b * (C >> 3) + (c >> 3)) * 2) * HW) * 8 + (c & 7);
        for (int i = threadIdx.x; i < HW; i += 256) {
            const float v = act_fn((p[i] - mean) * rstd, act);
            _Float16 h, l;
            sh16_split(v, scale, h, l);
            oh[(long long)i * 8] = h;
            oh[((long long)HW + i) * 8] = l;
        }
        return;
    }
    for (int i = threadIdx.x; i < HW; i += 256) p[i] = act_fn((p[i] - mean) * rstd, act);
}

// Large planes (Zencoder at 256^2 / 512^2: one plane = 0.25-1 MB, there are only B*C = 256-2048 of them): 1024 threads per
// plane and float4 accesses put 16x more bytes in flight per plane than the kernel above, which is what a one-block-per-plane
// reduction needs to approach HBM speed; passes 2 and 3 re-read the plane from L2.
__device__ __forceinline__ float block_sum1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i];
    return t;
}

__global__ __launch_bounds__(1024) void instnorm_act_wide_kernel(float* __restrict__ x, int HW, float eps, int act,
                                                                 _Float16* __restrict__ sh16, int C, float scale) {
    sh16_mode_on();
    __shared__ float red[16];
    float4* p4 = reinterpret_cast<float4*>(x + (long long)blockIdx.x * HW);
    const int n4 = HW >> 2;
    float s = 0.f;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = p4[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = block_sum1024(s, red) / HW;
    float q = 0.f;
    for (int i = threadIdx.x; i < n4; i += 1024) {
        const float4 v = p4[i];
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float var = block_sum1024(q, red) / HW;
    const float rstd = 1.f / sqrtf(var + eps);
    if (sh16) {
        const int b = blockIdx.x / C, c = blockIdx.x % C;
        _Float16* oh = sh16 + ((((long long)b * (C >> 3) + (c >> 3)) * 2) * HW) * 8 + (c & 7);
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 v4 = p4[i];
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = act_fn((vv[e] - mean) * rstd, act);
                _Float16 h, l;
                sh16_split(v, scale, h, l);
                oh[(long long)(i * 4 + e) * 8] = h;
                oh[((long long)HW + i * 4 + e) * 8] = l;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < n4; i += 1024) {
        float4 v = p4[i];
        v.x = act_fn((v.x - mean) * rstd, act); v.y = act_fn((v.y - mean) * rstd, act);
        v.z = act_fn((v.z - mean) * rstd, act); v.w = act_fn((v.w - mean) * rstd, act);
        p4[i] = v;
    }
}

// SH16 output (feeds the f16x3 conv): one block per (sample, group of 8 channels) so that every lane writes whole 16-byte
// units (8 channels of one pixel, hi plane and lo plane) instead of 2-byte elements 16 bytes apart from 8 different blocks.
__global__ __launch_bounds__(1024) void instnorm_act_sh16_kernel(const float* __restrict__ x, int HW, float eps, int act,
                                                                 uint4* __restrict__ sh16, int C, float scale) {
    sh16_mode_on();
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    __shared__ float red[16];
    __shared__ float mean_s[8], rstd_s[8];
    const int G = C >> 3, b = blockIdx.x / G, g = blockIdx.x % G;
    const float* base = x + ((long long)b * C + g * 8) * HW;
    const int n4 = HW >> 2;
    for (int c = 0; c < 8; ++c) {
        const float4* p4 = reinterpret_cast<const float4*>(base + (long long)c * HW);
        float s = 0.f;
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 v = p4[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
        const float mean = block_sum1024(s, red) / HW;
        float q = 0.f;
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 v = p4[i];
            const float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, d = v.w - mean;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
        const float var = block_sum1024(q, red) / HW;
        if (threadIdx.x == 0) {
            mean_s[c] = mean;
            rstd_s[c] = 1.f / sqrtf(var + eps);
        }
    }
    __syncthreads();
    uint4* oh = sh16 + ((long long)b * G + g) * 2 * HW;
    for (int i = threadIdx.x; i < HW; i += 1024) {
        half8v vh, vl;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = act_fn((base[(long long)c * HW + i] - mean_s[c]) * rstd_s[c], act);
            _Float16 h, l;
            sh16_split(v, scale, h, l);
            vh[c] = h;
            vl[c] = l;
        }
        oh[i] = __builtin_bit_cast(uint4, vh);
        oh[HW + i] = __builtin_bit_cast(uint4, vl);
    }
}

// Same, input in the C4 layout [B][C/4][HW][4] (output of an f16x3 conv): a block's 8 channels are two float4 per pixel.
__global__ __launch_bounds__(1024) void instnorm_c4_sh16_kernel(const float4* __restrict__ x, int HW, float eps, int act,
                                                                uint4* __restrict__ sh16, int C, float scale) {
    sh16_mode_on();
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    __shared__ float red[16];
    __shared__ float mean_s[8], rstd_s[8];
    const int G = C >> 3, b = blockIdx.x / G, g = blockIdx.x % G;
    const float4* p0 = x + ((long long)b * (C >> 2) + g * 2) * HW;
    const float4* p1 = p0 + HW;
    // statistics in ONE pass over the plane (it does not fit any cache: a second pass is another HBM read), as shifted sums
    // s = sum(x - K), q = sum((x - K)^2) with K = the channel's first element: mean = K + s/n, var = q/n - (s/n)^2 -- the
    // shift keeps |s/n| at the scale of the deviations, so the subtraction does not cancel
    const float4 ka = p0[0], kc = p1[0];
    const float K[8] = {ka.x, ka.y, ka.z, ka.w, kc.x, kc.y, kc.z, kc.w};
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < HW; i += 1024) {
        const float4 a = p0[i], c = p1[i];
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[e] - K[e];
            s[e] += d;
            q[e] += d * d;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float sm = block_sum1024(s[c], red) / HW;
        const float var = fmaxf(block_sum1024(q[c], red) / HW - sm * sm, 0.f);
        if (threadIdx.x == 0) {
            mean_s[c] = K[c] + sm;
            rstd_s[c] = 1.f / sqrtf(var + eps);
        }
    }
    __syncthreads();
    uint4* oh = sh16 + ((long long)b * G + g) * 2 * HW;
    for (int i = threadIdx.x; i < HW; i += 1024) {
        const float4 a = p0[i], c = p1[i];
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        half8v vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float o = act_fn((v[e] - mean_s[e]) * rstd_s[e], act);
            _Float16 h, l;
            sh16_split(o, scale, h, l);
            vh[e] = h;
            vl[e] = l;
        }
        oh[i] = __builtin_bit_cast(uint4, vh);
        oh[HW + i] = __builtin_bit_cast(uint4, vl);
    }
}

// Few (sample, channel group) pairs and large planes (Zencoder, 32 / 64 channels at 512^2 / 256^2): one block per pair leaves
// most CUs idle, so each pair's pixels are cut into NS slices -- kernel 1: per-slice (mean, M2) of the 8 channels (two-pass
// inside the slice); kernel 2: every block merges the NS partials of its pair (Chan et al., fixed order) and normalises,
// activates and splits its own slice.  stats: [pairs][NS][8][2] floats.
template <bool IN_C4>
__device__ __forceinline__ void instnorm_load8(const float* __restrict__ x, long long b, int g, int C, int HW, int i, float (&v)[8]) {
    if (IN_C4) {
        const float4* p0 = reinterpret_cast<const float4*>(x) + (b * (C >> 2) + g * 2) * HW;
        const float4 a = p0[i], c = p0[HW + i];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
        const float* base = x + (b * C + g * 8) * HW + i;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = base[(long long)c * HW];
    }
}
template <bool IN_C4>
__global__ __launch_bounds__(1024) void instnorm_slice_stats_kernel(const float* __restrict__ x, int HW, int C, int per,
                                                                    float* __restrict__ stats) {
    __shared__ float red[16];
    const int G = C >> 3, b = blockIdx.x / G, g = blockIdx.x % G, k = blockIdx.y, NS = gridDim.y;
    const int lo = k * per, hi = lo + per < HW ? lo + per : HW;
    // one pass: shifted sums around the slice's first pixel (see instnorm_c4_sh16_kernel)
    float K[8];
    instnorm_load8<IN_C4>(x, b, g, C, HW, lo, K);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = lo + threadIdx.x; i < hi; i += 1024) {
        float v[8];
        instnorm_load8<IN_C4>(x, b, g, C, HW, i, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[e] - K[e];
            s[e] += d;
            q[e] += d * d;
        }
    }
    const float cnt = (float)(hi - lo);
    float* o = stats + ((long long)blockIdx.x * NS + k) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float st = block_sum1024(s[e], red), qt = block_sum1024(q[e], red);
        if (threadIdx.x == 0) {
            o[e * 2] = K[e] + st / cnt;                          // slice mean
            o[e * 2 + 1] = fmaxf(qt - st * st / cnt, 0.f);       // slice M2 = sum (x - mean)^2
        }
    }
}
template <bool IN_C4>
__global__ __launch_bounds__(1024) void instnorm_slice_apply_kernel(const float* __restrict__ x, int HW, int C, int per, float eps,
                                                                    int act, const float* __restrict__ stats,
                                                                    uint4* __restrict__ sh16, float scale) {
    sh16_mode_on();
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    __shared__ float mean_s[8], rstd_s[8];
    const int G = C >> 3, b = blockIdx.x / G, g = blockIdx.x % G, k = blockIdx.y, NS = gridDim.y;
    if (threadIdx.x < 8) {
        const int e = threadIdx.x;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int j = 0; j < NS; ++j) {
            const int lo = j * per, hi = lo + per < HW ? lo + per : HW;
            const float nb = (float)(hi - lo);
            const float* o = stats + ((long long)blockIdx.x * NS + j) * 16 + e * 2;
            const float nn = n + nb, d = o[0] - mean;
            mean += d * nb / nn;
            m2 += o[1] + d * d * n * nb / nn;
            n = nn;
        }
        mean_s[e] = mean;
        rstd_s[e] = 1.f / sqrtf(m2 / HW + eps);          // biased variance (nn.InstanceNorm2d)
    }
    __syncthreads();
    const int lo = k * per, hi = lo + per < HW ? lo + per : HW;
    uint4* oh = sh16 + ((long long)b * G + g) * 2 * HW;
    for (int i = lo + threadIdx.x; i < hi; i += 1024) {
        float v[8];
        instnorm_load8<IN_C4>(x, b, g, C, HW, i, v);
        half8v vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float o = act_fn((v[e] - mean_s[e]) * rstd_s[e], act);
            _Float16 h, l;
            sh16_split(o, scale, h, l);
            vh[e] = h;
            vl[e] = l;
        }
        oh[i] = __builtin_bit_cast(uint4, vh);
        oh[HW + i] = __builtin_bit_cast(uint4, vl);
    }
}
// returns true (and launches) when the sliced form applies: scratch given, few pairs, large planes
static bool instnorm_sliced(const float* x, bool in_c4, int B, int C, int HW, float eps, int act, void* sh16, float* scratch,
                            hipStream_t s) {
    const int pairs = B * (C >> 3);
    if (!scratch || pairs >= 128 || HW < 16384) return false;
    int NS = (512 + pairs - 1) / pairs;
    if (NS > HW / 4096) NS = HW / 4096;
    if (NS > 64) NS = 64;
    const int per = (HW + NS - 1) / NS;
    const float sc = instnorm_sh16_scale(HW);
    if (in_c4) {
        hipLaunchKernelGGL(instnorm_slice_stats_kernel<true>, dim3(pairs, NS), dim3(1024), 0, s, x, HW, C, per, scratch);
        hipLaunchKernelGGL(instnorm_slice_apply_kernel<true>, dim3(pairs, NS), dim3(1024), 0, s, x, HW, C, per, eps, act, scratch,
                           static_cast<uint4*>(sh16), sc);
    } else {
        hipLaunchKernelGGL(instnorm_slice_stats_kernel<false>, dim3(pairs, NS), dim3(1024), 0, s, x, HW, C, per, scratch);
        hipLaunchKernelGGL(instnorm_slice_apply_kernel<false>, dim3(pairs, NS), dim3(1024), 0, s, x, HW, C, per, eps, act, scratch,
                           static_cast<uint4*>(sh16), sc);
    }
    return true;
}

// |(x - mean) / sqrt(var + eps)| <= sqrt(HW - 1) for any plane, and the activations used here (none / leaky / relu) do not
// increase magnitudes: the scale below can never saturate
float instnorm_sh16_scale(int HW) { return sh16_scale_for_bound(sqrtf((float)HW)); }

hipError_t instnorm_c4_to_sh16(const float* x_c4, int B, int C, int HW, float eps, int act, void* sh16, hipStream_t s,
                               float* scratch) {
    if (C & 7) return hipErrorInvalidValue;
    if (instnorm_sliced(x_c4, true, B, C, HW, eps, act, sh16, scratch, s)) return hipGetLastError();
    hipLaunchKernelGGL(instnorm_c4_sh16_kernel, dim3(B * (C >> 3)), dim3(1024), 0, s, reinterpret_cast<const float4*>(x_c4), HW,
                       eps, act, static_cast<uint4*>(sh16), C, instnorm_sh16_scale(HW));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// ConvTranspose2d(k3, s2, p1, op1) as four phase GEMMs (sean_model.cpp, Zencoder model.10; architecture.py:167-170):
//   out[2y+py][2x+px] = sum over dy <= py, dx <= px of Wt[:, :, py+1-2dy, px+1-2dx]^T x[y+dy][x+dx]        (x = 0 outside the image)
// convt_shift4: the four shifted views of x as planes of ONE buffer, xs[b][s][c][y][x] = x[b][c][y+dy][x+dx] with the shift order
// s = (0,1) | (0,0) | (1,0) | (1,1), so that every phase reads a contiguous range of planes: (0,0): s1; (0,1): s0..s1; (1,0): s1..s2;
// (1,1): s0..s3.  One thread = four pixels of a row.
__global__ __launch_bounds__(256) void convt_shift4_kernel(const float* __restrict__ x, float* __restrict__ xs, int C, int H, int W, long long n4) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n4) return;
    const int W4 = W >> 2;
    const int x4 = (int)(i % W4), y = (int)((i / W4) % H);
    const long long pc = i / ((long long)W4 * H);             // plane b * C + c
    const int c = (int)(pc % C);
    const long long b = pc / C;
    const long long HW = (long long)H * W;
    const float* r0 = x + pc * HW + (long long)y * W + 4 * x4;
    const float4 a = *reinterpret_cast<const float4*>(r0);
    const float an = 4 * x4 + 4 < W ? r0[4] : 0.f;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    float dn = 0.f;
    if (y + 1 < H) {
        d = *reinterpret_cast<const float4*>(r0 + W);
        dn = 4 * x4 + 4 < W ? r0[W + 4] : 0.f;
    }
    float* o = xs + ((b * 4) * C + c) * HW + (long long)y * W + 4 * x4;
    const long long ss = (long long)C * HW;
    *reinterpret_cast<float4*>(o) = make_float4(a.y, a.z, a.w, an);                 // (0, 1)
    *reinterpret_cast<float4*>(o + ss) = a;                                         // (0, 0)
    *reinterpret_cast<float4*>(o + 2 * ss) = d;                                     // (1, 0)
    *reinterpret_cast<float4*>(o + 3 * ss) = make_float4(d.y, d.z, d.w, dn);        // (1, 1)
}
hipError_t convt_shift4(const float* x, float* xs, int B, int C, int H, int W, hipStream_t s) {
    if (W & 3) return hipErrorInvalidValue;
    const long long n4 = (long long)B * C * H * (W >> 2);
    hipLaunchKernelGGL(convt_shift4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, xs, C, H, W, n4);
    return hipGetLastError();
}

// InstanceNorm2d(affine=False) + activation over the plane (b, c) of the ConvTranspose output, read from the four phase planes
// t[phase = 2 py + px][b][c][H][W] (+ bias[c]) and written depth-to-space: out[b][c][2y+py][2x+px].  Same three passes as
// instnorm_act_wide_kernel (mean, biased variance, write); one thread of the last pass = four output pixels of a row.
__global__ __launch_bounds__(1024) void instnorm_act_d2s_kernel(const float* __restrict__ t, const float* __restrict__ bias, float* __restrict__ out,
                                                                int BC, int C, int H, int W, float eps, int act) {
    __shared__ float red[16];
    const int HW = H * W, n4 = HW >> 2;
    const float bs = bias ? bias[blockIdx.x % C] : 0.f;
    const float4* p[4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) p[ph] = reinterpret_cast<const float4*>(t + ((long long)ph * BC + blockIdx.x) * HW);
    float s = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 v = p[ph][i];
            s += ((v.x + bs) + (v.y + bs)) + ((v.z + bs) + (v.w + bs));
        }
    const float mean = block_sum1024(s, red) / (4.f * HW);
    float q = 0.f;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 v = p[ph][i];
            const float a = v.x + bs - mean, b = v.y + bs - mean, c = v.z + bs - mean, d = v.w + bs - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    const float var = block_sum1024(q, red) / (4.f * HW);
    const float rstd = 1.f / sqrtf(var + eps);
    float* o = out + (long long)blockIdx.x * 4 * HW;
    const int W2 = W >> 1;                                       // pairs of input pixels per row = float4 units per output row
    for (int i = threadIdx.x; i < 2 * H * W2; i += 1024) {       // (output row, unit): row 2 y + py, columns 4 u .. 4 u + 3 = input x = 2 u, 2 u + 1
        const int u = i % W2, oy = i / W2, y = oy >> 1, py = oy & 1;
        const float2 e = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p[2 * py]) + y * W + 2 * u);        // px = 0
        const float2 f = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p[2 * py + 1]) + y * W + 2 * u);    // px = 1
        float4 v;
        v.x = act_fn((e.x + bs - mean) * rstd, act); v.y = act_fn((f.x + bs - mean) * rstd, act);
        v.z = act_fn((e.y + bs - mean) * rstd, act); v.w = act_fn((f.y + bs - mean) * rstd, act);
        *reinterpret_cast<float4*>(o + (long long)oy * (2 * W) + 4 * u) = v;
    }
}
hipError_t instnorm_act_d2s(const float* t, const float* bias, float* out, int B, int C, int H, int W, float eps, int act, hipStream_t s) {
    if ((W & 3) || ((H * W) & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(instnorm_act_d2s_kernel, dim3(B * C), dim3(1024), 0, s, t, bias, out, B * C, C, H, W, eps, act);
    return hipGetLastError();
}

hipError_t instnorm_act(float* x, int planes, int HW, float eps, int act, hipStream_t s, void* sh16, int C, float* scratch) {
    const float sc = instnorm_sh16_scale(HW);
    if (sh16 && (C & 7) == 0 && instnorm_sliced(x, false, planes / C, C, HW, eps, act, sh16, scratch, s)) return hipGetLastError();
    if (sh16 && (C & 7) == 0 && (HW & 3) == 0 && HW >= 4096) {
        hipLaunchKernelGGL(instnorm_act_sh16_kernel, dim3(planes / 8), dim3(1024), 0, s, x, HW, eps, act, static_cast<uint4*>(sh16), C, sc);
        return hipGetLastError();
    }
    if (HW >= 16384 && (HW & 3) == 0)
        hipLaunchKernelGGL(instnorm_act_wide_kernel, dim3(planes), dim3(1024), 0, s, x, HW, eps, act, static_cast<_Float16*>(sh16), C, sc);
    else
        hipLaunchKernelGGL(instnorm_act_kernel, dim3(planes), dim3(256), 0, s, x, HW, eps, act, static_cast<_Float16*>(sh16), C, sc);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// The custom LayerNorm of my_torchlib/module.py:189-205: per-sample mean and *unbiased* std over C*H*W,
// y = (x-mean)/(std+eps) * gamma[c] + beta[c]; then activation.  Two kernels: (1) per-block partial
// (count, mean, M2) with a block-local two-pass; (2) every block merges the partials (Chan et al.) and
// normalises its slice.  part: [B][nblk][3].
__global__ __launch_bounds__(256) void ln_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                         long long N, int nblk) {
    // N % 4 == 0 and `per` is a multiple of 4: float4 loads, several in flight per thread (the scalar one-load-per-iteration
    // version was latency bound: ~20 us for a 16k-element slice); the second pass re-reads the slice from L1 / L2
    __shared__ float red[4];
    const int b = blockIdx.y, k = blockIdx.x;
    const long long per = (((N + nblk - 1) / nblk) + 3) & ~3LL, lo = k * per, hi = (lo + per < N) ? lo + per : N;
    const float4* p4 = reinterpret_cast<const float4*>(x + (long long)b * N);
    const long long lo4 = lo >> 2, hi4 = hi >> 2;
    float s = 0.f;
#pragma unroll 4
    for (long long i = lo4 + threadIdx.x; i < hi4; i += 256) {
        const float4 v = p4[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float cnt = (float)(hi > lo ? hi - lo : 0);
    const float mean = cnt > 0 ? block_sum(s, red) / cnt : 0.f;
    float q = 0.f;
#pragma unroll 4
    for (long long i = lo4 + threadIdx.x; i < hi4; i += 256) {
        const float4 v = p4[i];
        const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
        q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float m2 = block_sum(q, red);
    if (threadIdx.x == 0) {
        float* o = part + ((long long)b * nblk + k) * 3;
        o[0] = cnt;
        o[1] = mean;
        o[2] = m2;
    }
}
// Merge of the (count, mean, M2) partials of one sample by the first wave of a block (Chan et al. pairwise, butterfly over
// the lanes: a fixed association, so results are reproducible); every thread of the block receives (mean, 1 / (std + eps)),
// std unbiased like the reference's x.view(B, -1).std(1) (my_torchlib/module.py:192-196).  nblk <= 128.
__device__ __forceinline__ void ln_merge_partials(const float* __restrict__ part, int nblk, float eps, float& mean_out,
                                                  float& inv_out) {
    __shared__ float s_stat[2];
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int k = lane; k < nblk; k += 64) {
            const float nb = part[k * 3 + 0], mb = part[k * 3 + 1], qb = part[k * 3 + 2];
            if (nb > 0.f) {
                const float nn = n + nb, d = mb - mean;
                mean += d * nb / nn;
                m2 += qb + d * d * n * nb / nn;
                n = nn;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float nb = __shfl_xor(n, off, 64), mb = __shfl_xor(mean, off, 64), qb = __shfl_xor(m2, off, 64);
            const float nn = n + nb;
            if (nn > 0.f) {
                // symmetric form (both partners must arrive at the same merged triple)
                const float d = mb - mean;
                const float merged_mean = (n * mean + nb * mb) / nn;
                m2 = m2 + qb + d * d * n * nb / nn;
                mean = merged_mean;
                n = nn;
            }
        }
        if (lane == 0) {
            s_stat[0] = mean;
            s_stat[1] = 1.f / (sqrtf(m2 / (n - 1.f)) + eps);
        }
    }
    __syncthreads();
    mean_out = s_stat[0];
    inv_out = s_stat[1];
}

__global__ __launch_bounds__(256) void ln_apply_kernel(float* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       long long N, int HW, int nblk, float eps, int act) {
    const int b = blockIdx.y;
    float mean, inv;
    ln_merge_partials(part + (long long)b * nblk * 3, nblk, eps, mean, inv);
    float* p = x + (long long)b * N;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const int c = (int)(i / HW);
        p[i] = act_fn((p[i] - mean) * inv * gamma[c] + beta[c], act);
    }
}
// Same normalisation with layout conversion (shape decoder on the f16x3 conv path): input NCHW or C4 ([B][C/4][HW][4], the
// f16x3 convs' output), output SH16 ([B][C/8][hi|lo][HW][8] f16, scaled by out_scale: feeds the next f16x3 conv) or NCHW f32.
// One thread = 8 channels of one pixel.  The statistics (ln_partial_kernel) do not depend on the layout: a sample is one
// contiguous block of C*HW floats either way.
template <bool IN_C4, bool OUT_SH16>
__global__ __launch_bounds__(256) void ln_apply_conv_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            void* __restrict__ out, int C, int HW, int nblk, float eps, int act,
                                                            float out_scale) {
    typedef _Float16 half8v __attribute__((ext_vector_type(8)));
    sh16_mode_on();
    const int b = blockIdx.y;
    float mean, inv;
    ln_merge_partials(part + (long long)b * nblk * 3, nblk, eps, mean, inv);
    const int G = C >> 3;
    const long long items = (long long)G * HW;
    const float* xb = x + (long long)b * C * HW;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < items; i += (long long)gridDim.x * 256) {
        const int g = (int)(i / HW), p = (int)(i % HW);
        float v[8];
        if (IN_C4) {
            const float4 a = reinterpret_cast<const float4*>(xb)[(long long)(2 * g) * HW + p];
            const float4 c = reinterpret_cast<const float4*>(xb)[(long long)(2 * g + 1) * HW + p];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xb[(long long)(g * 8 + e) * HW + p];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_fn((v[e] - mean) * inv * gamma[g * 8 + e] + beta[g * 8 + e], act);
        if (OUT_SH16) {
            half8v vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h, l;
                sh16_split(v[e], out_scale, h, l);
                vh[e] = h;
                vl[e] = l;
            }
            uint4* o = static_cast<uint4*>(out) + ((long long)b * G + g) * 2 * HW;
            o[p] = __builtin_bit_cast(uint4, vh);
            o[HW + p] = __builtin_bit_cast(uint4, vl);
        } else {
            float* o = static_cast<float*>(out) + (long long)b * C * HW;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[(long long)(g * 8 + e) * HW + p] = v[e];
        }
    }
}

hipError_t layernorm_act_conv(const float* x, int in_c4, void* out, int out_sh16, float out_scale, const float* gamma,
                              const float* beta, float* part, int B, int C, int HW, float eps, int act, hipStream_t s) {
    if (C & 7) return hipErrorInvalidValue;
    const long long N = (long long)C * HW;      // a multiple of 8
    int nblk = (int)((N + 4095) / 4096);
    if (nblk > 128) nblk = 128;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(ln_partial_kernel, dim3(nblk, B), dim3(256), 0, s, x, part, N, nblk);
    int gx = (int)((N / 8 + 255) / 256);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
#define LN_LAUNCH(A, O) hipLaunchKernelGGL((ln_apply_conv_kernel<A, O>), dim3(gx, B), dim3(256), 0, s, x, part, gamma, beta, out, C, HW, nblk, eps, act, out_scale)
    if (in_c4 && out_sh16) LN_LAUNCH(true, true);
    else if (in_c4) LN_LAUNCH(true, false);
    else if (out_sh16) LN_LAUNCH(false, true);
    else LN_LAUNCH(false, false);
#undef LN_LAUNCH
    return hipGetLastError();
}

hipError_t layernorm_act(float* x, const float* gamma, const float* beta, float* part, int B, int C, int HW, float eps,
                         int act, hipStream_t s) {
    const long long N = (long long)C * HW;
    if (N & 3) return hipErrorInvalidValue;      // ln_partial_kernel reads float4
    int nblk = (int)((N + 4095) / 4096);
    if (nblk > 128) nblk = 128;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(ln_partial_kernel, dim3(nblk, B), dim3(256), 0, s, x, part, N, nblk);
    int gx = (int)((N + 2047) / 2048);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ln_apply_kernel, dim3(gx, B), dim3(256), 0, s, x, part, gamma, beta, N, HW, nblk, eps, act);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Zencoder region average pooling (architecture.py:179-205): codes [B,F,h,w], labels nearest-down-sampled to
// (h,w) from the full-res label map [B,S,S]; out[b,j,f] = mean of codes[b,f,p] over pixels with label j
// (0 if the region is absent).  One block per (f, b); 19 register accumulators per thread.
__global__ __launch_bounds__(256) void region_mean_kernel(const float* __restrict__ codes,
                                                          const uint8_t* __restrict__ lab, float* __restrict__ out,
                                                          int F, int h, int w, int S, int c4) {
    __shared__ float red[4];
    const int f = blockIdx.x, b = blockIdx.y;
    const int fy = S / h, fx = S / w;
    // c4: codes in the C4 layout [B][F/4][h*w][4] (output of the f16x3 conv), else NCHW
    const float* p = c4 ? codes + ((long long)b * (F >> 2) + (f >> 2)) * h * w * 4 + (f & 3)
                        : codes + ((long long)b * F + f) * h * w;
    const int ps = c4 ? 4 : 1;
    const uint8_t* lb = lab + (long long)b * S * S;
    float acc[19], cnt[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) { acc[j] = 0.f; cnt[j] = 0.f; }
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const int y = i / w, x = i % w;
        const int l = lb[(long long)(y * fy) * S + x * fx];
        const float v = p[(long long)i * ps];
#pragma unroll
        for (int j = 0; j < 19; ++j)
            if (l == j) { acc[j] += v; cnt[j] += 1.f; }
    }
#pragma unroll
    for (int j = 0; j < 19; ++j) {
        const float sv = block_sum(acc[j], red);
        const float cv = block_sum(cnt[j], red);
        if (threadIdx.x == 0) out[((long long)b * 19 + j) * F + f] = cv > 0.f ? sv / cv : 0.f;
    }
}
// Planar input, run-length form (exact-f32 path): a thread's pixels i, i + 256, ... lie in one image column when w divides 256, where
// labels change rarely: it keeps a running (label, sum, count) and adds it to ITS OWN per-label LDS slot only when the label changes
// (the 19-way compare-and-add per element of the kernel above made it VALU-bound: 0.66 ms for the 1.07 GB feature map of 8 images at
// 512^2).  Deterministic: private slots, then the same block-wide tree per label.
__global__ __launch_bounds__(256) void region_mean_runs_kernel(const float* __restrict__ codes, const uint8_t* __restrict__ lab,
                                                               float* __restrict__ out, int F, int h, int w, int S) {
    __shared__ float acc[19][256], cnt[19][256];
    __shared__ float red[4];
    const int f = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
    const int fy = S / h, fx = S / w;
    const float* p = codes + ((long long)b * F + f) * h * w;
    const uint8_t* lb = lab + (long long)b * S * S;
#pragma unroll
    for (int j = 0; j < 19; ++j) { acc[j][t] = 0.f; cnt[j][t] = 0.f; }
    int cur = -1;
    float sum = 0.f, n = 0.f;
    const int hw = h * w;
    for (int i0 = t; i0 < hw; i0 += 8 * 256) {            // eight loads in flight per thread, consumed in pixel order
        float v[8];
        int l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 256;
            l[k] = -1;
            v[k] = 0.f;
            if (i < hw) {
                const int y = i / w, x = i - y * w;
                l[k] = lb[(long long)(y * fy) * S + x * fx];
                v[k] = p[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (l[k] < 0) continue;                        // past the end
            if (l[k] != cur) {
                if (cur >= 0 && cur < 19) { acc[cur][t] += sum; cnt[cur][t] += n; }
                cur = l[k];
                sum = 0.f;
                n = 0.f;
            }
            sum += v[k];
            n += 1.f;
        }
    }
    if (cur >= 0 && cur < 19) { acc[cur][t] += sum; cnt[cur][t] += n; }
    for (int j = 0; j < 19; ++j) {
        const float sv = block_sum(acc[j][t], red);
        const float cv = block_sum(cnt[j][t], red);
        if (t == 0) out[((long long)b * 19 + j) * F + f] = cv > 0.f ? sv / cv : 0.f;
    }
}
// C4 input [B][F/4][h*w][4]: one block per (4-channel group, sample), float4 loads.  A thread walks down a column-ish
// sequence of pixels (stride 256), where labels change rarely: it keeps a running (label, sum4, count) and flushes it to the
// per-label LDS accumulators only when the label changes.  (LDS float atomics: the summation order, hence the last bits of a
// code, can differ from run to run -- ~1e-7 relative, far inside the 1e-3 bar.)
__global__ __launch_bounds__(256) void region_mean_c4_kernel(const float4* __restrict__ codes, const uint8_t* __restrict__ lab,
                                                             float* __restrict__ out, int F, int h, int w, int S) {
    __shared__ float acc[19][4];
    __shared__ float cnt[19];
    const int fg = blockIdx.x, b = blockIdx.y;
    if (threadIdx.x < 19 * 4) acc[threadIdx.x >> 2][threadIdx.x & 3] = 0.f;
    if (threadIdx.x < 19) cnt[threadIdx.x] = 0.f;
    __syncthreads();
    const float4* p = codes + ((long long)b * (F >> 2) + fg) * h * w;
    const uint8_t* lb = lab + (long long)b * S * S;
    const int fy = S / h, fx = S / w;
    int cur = -1;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float n = 0.f;
    auto flush = [&]() {
        if (cur >= 0 && cur < 19) {
            atomicAdd(&acc[cur][0], sum.x); atomicAdd(&acc[cur][1], sum.y);
            atomicAdd(&acc[cur][2], sum.z); atomicAdd(&acc[cur][3], sum.w);
            atomicAdd(&cnt[cur], n);
        }
    };
    for (int i = threadIdx.x; i < h * w; i += 256) {
        const int y = i / w, x = i % w;
        const int l = lb[(long long)(y * fy) * S + x * fx];
        const float4 v = p[i];
        if (l != cur) {
            flush();
            cur = l;
            sum = make_float4(0.f, 0.f, 0.f, 0.f);
            n = 0.f;
        }
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        n += 1.f;
    }
    flush();
    __syncthreads();
    if (threadIdx.x < 19 * 4) {
        const int j = threadIdx.x >> 2, c = threadIdx.x & 3;
        out[((long long)b * 19 + j) * F + fg * 4 + c] = cnt[j] > 0.f ? acc[j][c] / cnt[j] : 0.f;
    }
}

hipError_t region_mean(const float* codes, const uint8_t* lab, float* out, int B, int F, int h, int w, int S,
                       hipStream_t s, int c4) {
    if (c4 && (F & 3) == 0) {
        hipLaunchKernelGGL(region_mean_c4_kernel, dim3(F >> 2, B), dim3(256), 0, s, reinterpret_cast<const float4*>(codes), lab,
                           out, F, h, w, S);
        return hipGetLastError();
    }
    if (!c4 && h * w >= 4096) {
        hipLaunchKernelGGL(region_mean_runs_kernel, dim3(F, B), dim3(256), 0, s, codes, lab, out, F, h, w, S);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(region_mean_kernel, dim3(F, B), dim3(256), 0, s, codes, lab, out, F, h, w, S, c4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// nn.MaxPool2d(3, stride 2, padding 1) (resnet.py:63)
__global__ void maxpool3x3s2_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int H,
                                    int W, int Ho, int Wo) {
    const long long n = planes * Ho * Wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const long long pl = i / ((long long)Wo * Ho);
        const float* p = in + pl * H * W;
        float m = -3.4e38f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = 2 * y + dy, xx = 2 * x + dx;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) m = fmaxf(m, p[yy * W + xx]);
            }
        out[i] = m;
    }
}
hipError_t maxpool3x3s2(const float* in, float* out, long long planes, int H, int W, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long n = planes * Ho * Wo;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid), dim3(256), 0, s, in, out, planes, H, W, Ho, Wo);
    return hipGetLastError();
}

// F.avg_pool2d(x, x.size()[2:]) : one wave per plane
__global__ __launch_bounds__(256) void gap_kernel(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                  int HW) {
    const int pl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pl >= planes) return;
    const float* p = in + (long long)pl * HW;
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += p[i];
    s = wave_sum(s);
    if (lane == 0) out[pl] = s / HW;
}
hipError_t global_avg_pool(const float* in, float* out, int planes, int HW, hipStream_t s) {
    hipLaunchKernelGGL(gap_kernel, dim3((planes + 3) / 4), dim3(256), 0, s, in, out, planes, HW);
    return hipGetLastError();
}

// out[pl,p] = in[pl,p] * (sc[pl] + sc_add) + (sh ? sh[pl] : 0) + (other ? other[pl,p] : 0)
__global__ void chan_affine_kernel(const float* __restrict__ in, const float* __restrict__ sc, float sc_add,
                                   const float* __restrict__ sh, const float* __restrict__ other,
                                   float* __restrict__ out, long long planes, int HW) {
    const long long n = planes * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long pl = i / HW;
        float v = in[i] * (sc[pl] + sc_add);
        if (sh) v += sh[pl];
        if (other) v += other[i];
        out[i] = v;
    }
}
hipError_t chan_affine(const float* in, const float* sc, float sc_add, const float* sh, const float* other, float* out,
                       long long planes, int HW, hipStream_t s) {
    const long long n = planes * HW;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(chan_affine_kernel, dim3(grid), dim3(256), 0, s, in, sc, sc_add, sh, other, out, planes, HW);
    return hipGetLastError();
}

// ---- C4-layout ([B][C/4][H][W][4] f32) variants for the f16x3 BiSeNet trunk.  Producers of tensors that feed an INC4 conv
// (conv_sh16.h) also record max |out| * SH16_ACT_SCALE in `amax` (sh16.h slot convention).
__device__ __forceinline__ void amax_commit(unsigned* slot, float amax) {      // whole block; see sh16_block_slot_max
    sh16_block_slot_max(slot, amax * SH16_ACT_SCALE);
}
// nn.MaxPool2d(3, 2, 1) (resnet.py:75): NCHW in -> C4 out
__global__ __launch_bounds__(1024) void maxpool3x3s2_c4_kernel(const float* __restrict__ in, float4* __restrict__ out, unsigned* amax_slot, int B, int C,
                                       int H, int W, int Ho, int Wo) {
    const long long n = (long long)B * (C >> 2) * Ho * Wo;
    float amax = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const long long bg = i / ((long long)Wo * Ho);            // b * C/4 + group
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* p = in + (bg * 4 + e) * H * W;
            float v = -3.4e38f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = 2 * y + dy, xx = 2 * x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = fmaxf(v, p[yy * W + xx]);
                }
            m[e] = v;
            amax = fmaxf(amax, fabsf(v));
        }
        out[i] = make_float4(m[0], m[1], m[2], m[3]);
    }
    amax_commit(amax_slot, amax);
}
hipError_t maxpool3x3s2_c4(const float* in, float* out, unsigned* amax, int B, int C, int H, int W, hipStream_t s) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long n = (long long)B * (C >> 2) * Ho * Wo;
    hipLaunchKernelGGL(maxpool3x3s2_c4_kernel, dim3(sh16_ew_grid(n)), dim3(SH16_EW_THREADS), 0, s, in, reinterpret_cast<float4*>(out), amax, B, C, H, W,
                       Ho, Wo);
    return hipGetLastError();
}
// F.avg_pool2d(x, x.size()[2:]) on a C4 tensor: one wave per (sample, 4-channel group) -> out[b*C + c]
__global__ __launch_bounds__(256) void gap_c4_kernel(const float4* __restrict__ in, float* __restrict__ out, int groups, int HW) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= groups) return;
    const float4* p = in + (long long)g * HW;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < HW; i += 64) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    a.x = wave_sum(a.x); a.y = wave_sum(a.y); a.z = wave_sum(a.z); a.w = wave_sum(a.w);
    if (lane == 0) *reinterpret_cast<float4*>(out + (long long)g * 4) = make_float4(a.x / HW, a.y / HW, a.z / HW, a.w / HW);
}
hipError_t global_avg_pool_c4(const float* in, float* out, int B, int C, int HW, hipStream_t s) {
    const int groups = B * (C >> 2);
    hipLaunchKernelGGL(gap_c4_kernel, dim3((groups + 3) / 4), dim3(256), 0, s, reinterpret_cast<const float4*>(in), out, groups, HW);
    return hipGetLastError();
}
// chan_affine on C4 tensors: out = in * (sc[b,c] + sc_add) + (sh ? sh[b,c] : 0) + (other ? other : 0)
__global__ __launch_bounds__(1024) void chan_affine_c4_kernel(const float4* __restrict__ in, const float* __restrict__ sc, float sc_add,
                                      const float* __restrict__ sh, const float4* __restrict__ other, float4* __restrict__ out,
                                      unsigned* amax_slot, long long groups, int HW) {
    const long long n = groups * HW;
    float amax = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long g = i / HW;
        const float4 a = *reinterpret_cast<const float4*>(sc + g * 4);
        float4 v = in[i];
        v.x *= a.x + sc_add; v.y *= a.y + sc_add; v.z *= a.z + sc_add; v.w *= a.w + sc_add;
        if (sh) {
            const float4 t = *reinterpret_cast<const float4*>(sh + g * 4);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (other) {
            const float4 t = other[i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        out[i] = v;
        amax = fmaxf(fmaxf(amax, fabsf(v.x)), fabsf(v.y));
        amax = fmaxf(fmaxf(amax, fabsf(v.z)), fabsf(v.w));
    }
    amax_commit(amax_slot, amax);
}
hipError_t chan_affine_c4(const float* in, const float* sc, float sc_add, const float* sh, const float* other, float* out,
                          unsigned* amax, int B, int C, int HW, hipStream_t s) {
    const long long groups = (long long)B * (C >> 2), n = groups * HW;
    hipLaunchKernelGGL(chan_affine_c4_kernel, dim3(sh16_ew_grid(n)), dim3(SH16_EW_THREADS), 0, s, reinterpret_cast<const float4*>(in), sc, sc_add, sh,
                       reinterpret_cast<const float4*>(other), reinterpret_cast<float4*>(out), amax, groups, HW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// ResNet-18 stem (resnet.py:61,72-73): conv 7x7 s2 p3, 3 -> 64, eval-BN folded into (w, bias), ReLU.
// Direct VALU conv (K = 147 is too short for the matrix cores): block = 16x16 output pixels, one pixel per thread, all 64
// output channels in registers; the 37x37x3 input patch in LDS (one read per tap); the weights, transposed on the host to
// [tap][64], arrive through wave-uniform scalar loads and enter the packed FMAs (v_pk_fma_f32) as SGPR operands -- no LDS
// traffic for them (the previous version read 16 weights from LDS per 16 FMAs and was LDS-issue bound: 0.48 ms -> see
// DESIGN.md for the measured time at B=8, 512x512).
typedef float stem_f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void stem7x7_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                      const float* __restrict__ bias, float* __restrict__ out, int H,
                                                      int W, int Ho, int Wo) {
    __shared__ float patch[3][37][38];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16, b = blockIdx.z;
    for (int e = threadIdx.x; e < 3 * 37 * 37; e += 256) {
        const int c = e / (37 * 37), r = e % (37 * 37), py = r / 37, px = r % 37;
        const int y = oy0 * 2 - 3 + py, x = ox0 * 2 - 3 + px;
        patch[c][py][px] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                               ? in[((long long)b * 3 + c) * H * W + (long long)y * W + x]
                               : 0.f;
    }
    __syncthreads();
    stem_f2 acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = stem_f2{0.f, 0.f};
#pragma unroll 1
    for (int cky = 0; cky < 21; ++cky) {                  // (channel, kernel row)
        const int c = cky / 7, ky = cky % 7;
        const float* prow = &patch[c][ty * 2 + ky][tx * 2];
        const stem_f2* wrow = reinterpret_cast<const stem_f2*>(wt + (size_t)cky * 7 * 64);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float v = prow[kx];
            const stem_f2 vv = stem_f2{v, v};
#pragma unroll
            for (int k = 0; k < 32; ++k) acc[k] += wrow[kx * 32 + k] * vv;
        }
    }
    const int ox = ox0 + tx, oy = oy0 + ty;
    if (ox < Wo && oy < Ho) {
        float* o = out + (((long long)b * 64) * Ho + oy) * Wo + ox;
        const long long cs = (long long)Ho * Wo;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float v0 = acc[k].x + bias[2 * k], v1 = acc[k].y + bias[2 * k + 1];
            o[(2 * k) * cs] = v0 > 0.f ? v0 : 0.f;
            o[(2 * k + 1) * cs] = v1 > 0.f ? v1 : 0.f;
        }
    }
}
// ReflectionPad2d(1) + Conv2d(3, Cout, 3) (Zencoder stem, architecture.py:158-160): three input channels are far too few
// for the matrix cores (the MFMA kernel pads K from 27 to 144); direct VALU conv, one pixel per thread, the 27 inputs in
// registers, weights through wave-uniform (scalar) loads.  HBM-write bound (Cout planes out, 3 planes in).
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_c3_reflect_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int B, int H, int W) {
    const long long HW = (long long)H * W;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= B * HW) return;
    const int b = (int)(i / HW), y = (int)((i % HW) / W), x = (int)(i % W);
    float v[27];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
        xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c * 9 + t] = in[((long long)b * 3 + c) * HW + (long long)yy * W + xx];
    }
    float* o = out + (long long)b * COUT * HW + (long long)y * W + x;
#pragma unroll 4
    for (int k = 0; k < COUT; ++k) {
        float a = bias[k];
#pragma unroll
        for (int j = 0; j < 27; ++j) a += w[k * 27 + j] * v[j];
        o[(long long)k * HW] = a;
    }
}

hipError_t conv3x3_c3_reflect(const float* in, const float* w, const float* bias, float* out, int B, int Cout, int H, int W,
                              hipStream_t s) {
    if (Cout != 32) return hipErrorInvalidValue;
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(conv3x3_c3_reflect_kernel<32>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, w, bias, out, B, H, W);
    return hipGetLastError();
}

hipError_t stem7x7(const float* in, const float* w, const float* bias, float* out, int B, int H, int W, hipStream_t s) {
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    hipLaunchKernelGGL(stem7x7_kernel, dim3((Wo + 15) / 16, (Ho + 15) / 16, B), dim3(256), 0, s, in, w, bias, out, H, W,
                       Ho, Wo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// BiSeNet tail (model.py:250 + my_parsing_util.py:45-54): bilinear align_corners=True up-sampling of the 19
// logit planes to (H,W), argmax over classes, remap BiSeNet ids -> CelebAMask-HQ ids with a 19-entry LUT.
// logits_out (optional) receives the up-sampled logits [B,19,H,W] for tests.
// c4 != 0: `lg` is the C4 tensor [B][5][h][w][4] (19 classes + one padding row) written by the f16x3 classifier conv.
__global__ void bilinear_argmax_kernel(const float* __restrict__ lg, uint8_t* __restrict__ out,
                                       float* __restrict__ logits_out, const uint8_t* __restrict__ remap, int B, int h,
                                       int w, int H, int W, int c4) {
    const long long n = (long long)B * H * W;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int b = (int)(i / ((long long)W * H));
        const float fy = sy * y, fx = sx * x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
        const float ly = fy - y0, lx = fx - x0;
        float best = -3.4e38f;
        int bi = 0;
        for (int c = 0; c < 19; ++c) {
            const float* p = c4 ? lg + ((long long)b * 5 + (c >> 2)) * h * w * 4 + (c & 3) : lg + ((long long)b * 19 + c) * h * w;
            const int es = c4 ? 4 : 1;                     // element stride between neighbouring pixels
            // same association as ATen's upsample_bilinear2d: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
            const float v = (1.f - ly) * ((1.f - lx) * p[(y0 * w + x0) * es] + lx * p[(y0 * w + x1) * es]) +
                            ly * ((1.f - lx) * p[(y1 * w + x0) * es] + lx * p[(y1 * w + x1) * es]);
            if (logits_out) logits_out[(((long long)b * 19 + c) * H + y) * W + x] = v;
            if (v > best) { best = v; bi = c; }
        }
        out[i] = remap ? remap[bi] : (uint8_t)bi;
    }
}
hipError_t bilinear_argmax(const float* lg, uint8_t* out, float* logits_out, const uint8_t* remap, int B, int h, int w,
                           int H, int W, hipStream_t s, int c4) {
    const long long n = (long long)B * H * W;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(bilinear_argmax_kernel, dim3(grid), dim3(256), 0, s, lg, out, logits_out, remap, B, h, w, H, W, c4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Shape decoder tail (shape_branch/model.py:184-187 + shape_util.py:17-20): insert the hair logit at class 13,
// softmax over 19, argmax -> uint8 label.  probs (optional) receives the softmax [B,19,H,W].
// c4 != 0: the logits are the C4 tensors the f16x3 output convs write (hair [B][1][HW][4], face [B][5][HW][4]; rows padded)
__global__ void shape_softmax_kernel(const float* __restrict__ hair, const float* __restrict__ face,
                                     uint8_t* __restrict__ lab, float* __restrict__ probs, int B, int HW, int c4) {
    const long long n = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        float v[19];
        if (c4) {
            float f[20];
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                const float4 t = reinterpret_cast<const float4*>(face)[((long long)b * 5 + g) * HW + p];
                f[g * 4] = t.x; f[g * 4 + 1] = t.y; f[g * 4 + 2] = t.z; f[g * 4 + 3] = t.w;
            }
#pragma unroll
            for (int c = 0; c < 19; ++c) v[c] = c == 13 ? hair[((long long)b * HW + p) * 4] : f[c < 13 ? c : c - 1];
        } else {
#pragma unroll
            for (int c = 0; c < 19; ++c)
                v[c] = c == 13 ? hair[(long long)b * HW + p]
                               : face[((long long)b * 18 + (c < 13 ? c : c - 1)) * HW + p];
        }
        float m = v[0];
        int bi = 0;
#pragma unroll
        for (int c = 1; c < 19; ++c)
            if (v[c] > m) { m = v[c]; bi = c; }
        lab[i] = (uint8_t)bi;
        if (probs) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 19; ++c) { v[c] = __expf(v[c] - m); s += v[c]; }
            const float inv = 1.f / s;
#pragma unroll
            for (int c = 0; c < 19; ++c) probs[((long long)b * 19 + c) * HW + p] = v[c] * inv;
        }
    }
}
hipError_t shape_softmax(const float* hair, const float* face, uint8_t* lab, float* probs, int B, int HW,
                         hipStream_t s, int c4) {
    const long long n = (long long)B * HW;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(shape_softmax_kernel, dim3(grid), dim3(256), 0, s, hair, face, lab, probs, B, HW, c4);
    return hipGetLastError();
}
// first C channels of a C4 tensor with Cpad channels -> NCHW [B][C][HW]
__global__ void c4_rows_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int Cpad, int HW) {
    const long long n = (long long)B * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((long long)HW * C));
        out[i] = in[(((long long)b * (Cpad >> 2) + (c >> 2)) * HW + p) * 4 + (c & 3)];
    }
}
hipError_t c4_rows_to_nchw(const float* in, float* out, int B, int C, int Cpad, int HW, hipStream_t s) {
    const long long n = (long long)B * C * HW;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(c4_rows_to_nchw_kernel, dim3(grid), dim3(256), 0, s, in, out, B, C, Cpad, HW);
    return hipGetLastError();
}

// Shape encoder inputs (ui/backend.py:81-83, shape_util.py:6-26, shape_branch/model.py:96-100):
// label map uint8 [B,HW] (255 = no class) -> hair_in [B,1+40,HW] (one-hot of class 13 + positional channels) and
// face_in [B,18+40,HW] (one-hot of the other 18 classes + positional channels).  pos: [40][HW].
__global__ void shape_inputs_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ pos,
                                    float* __restrict__ hair_in, float* __restrict__ face_in, int B, int HW) {
    const long long n = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        const int l = lab[i];
        float* h = hair_in + (long long)b * 41 * HW + p;
        float* f = face_in + (long long)b * 58 * HW + p;
        h[0] = l == 13 ? 1.f : 0.f;
        for (int c = 0; c < 18; ++c) f[(long long)c * HW] = (l == (c < 13 ? c : c + 1)) ? 1.f : 0.f;
        for (int k = 0; k < 40; ++k) {
            const float v = pos[(long long)k * HW + p];
            h[(long long)(1 + k) * HW] = v;
            f[(long long)(18 + k) * HW] = v;
        }
    }
}
// The same inputs as SH16 tensors (f16x3 shape encoder): channels padded to 48 / 64 with zeros, scale 2^14 (|value| <= 1)
__global__ void shape_inputs_sh16_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ pos,
                                         uint4* __restrict__ hair_in, uint4* __restrict__ face_in, int B, int HW, float scale) {
    sh16_mode_on();
    const long long n = (long long)B * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / HW), p = (int)(i % HW);
        const int l = lab[i];
        float pe[40];
#pragma unroll
        for (int k = 0; k < 40; ++k) pe[k] = pos[(long long)k * HW + p];
        auto put = [&](uint4* base, int G, int g, const float (&v)[8]) {
            half8 h, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 he, le;
                sh16_split(v[e], scale, he, le);
                h[e] = he;
                lo[e] = le;
            }
            base[(((long long)b * G + g) * 2 + 0) * HW + p] = __builtin_bit_cast(uint4, h);
            base[(((long long)b * G + g) * 2 + 1) * HW + p] = __builtin_bit_cast(uint4, lo);
        };
#pragma unroll
        for (int g = 0; g < 6; ++g) {                   // hair: [one-hot of class 13][40 positional][7 zeros]
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = g * 8 + e;
                v[e] = c == 0 ? (l == 13 ? 1.f : 0.f) : (c <= 40 ? pe[c - 1] : 0.f);
            }
            put(hair_in, 6, g, v);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {                   // face: [18 one-hot][40 positional][6 zeros]
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = g * 8 + e;
                v[e] = c < 18 ? ((l == (c < 13 ? c : c + 1)) ? 1.f : 0.f) : (c < 58 ? pe[c - 18] : 0.f);
            }
            put(face_in, 8, g, v);
        }
    }
}
hipError_t shape_inputs_sh16(const uint8_t* lab, const float* pos, void* hair_in, void* face_in, int B, int HW, float scale,
                             hipStream_t s) {
    const long long n = (long long)B * HW;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(shape_inputs_sh16_kernel, dim3(grid), dim3(256), 0, s, lab, pos, static_cast<uint4*>(hair_in),
                       static_cast<uint4*>(face_in), B, HW, scale);
    return hipGetLastError();
}
// First layer of BOTH shape encoders as a label table (exact-f32 path; shape_branch/model.py:74-79,96-100; shape_util.py:6-26).
// Its input is one-hot mask channels (hair: class 13; face: the 18 other classes) + 40 positional channels that never change:
//     conv4x4s2(cat[onehot, pos]) = posconst + sum over the 16 taps of  W[:, channel(label at the tap), tap]
// posconst[c][y][x] = bias + the positional channels' part (computed once at ch_finalize by the conv kernel itself on an all-'no class'
// label map), the tap sum is 16 table rows per output pixel instead of 2 x 58 x 16 products per output element: the same real number in
// another association of the f32 sum (1e-7 relative).  Taps outside the image and labels >= 19 select the all-zero row; label 13
// selects zero in the face half and the hair row in the hair half.
//   tab: [2 encoders][16 taps][20 rows][32 channels]   (row 19 = zeros)
// Block = 32 output pixels of a row x the 32 channels of one encoder; thread = (pixel, group of 4 channels): 16 ds_read_b128 + 4 stores.
__global__ __launch_bounds__(256) void shape_enc_l0_kernel(const uint8_t* __restrict__ lab, const float* __restrict__ tab,
                                                           const float* __restrict__ pc_hair, const float* __restrict__ pc_face,
                                                           float* __restrict__ out_hair, float* __restrict__ out_face, int B, int S) {
    __shared__ __attribute__((aligned(16))) float T[16 * 20 * 32];
    const int w = blockIdx.y;                     // 0 hair, 1 face
    float* out = w == 0 ? out_hair : out_face;
    if (!out) return;
    const float* pc = w == 0 ? pc_hair : pc_face;
    for (int i = threadIdx.x; i < 16 * 20 * 32 / 4; i += 256)
        reinterpret_cast<float4*>(T)[i] = reinterpret_cast<const float4*>(tab + (size_t)w * 16 * 20 * 32)[i];
    __syncthreads();
    const int So = S >> 1, tpr = So >> 5;         // tiles of 32 pixels per output row
    const int px = threadIdx.x & 31, cg = threadIdx.x >> 5;
    const long long ntile = (long long)B * So * tpr;
    for (long long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int b = (int)(t / ((long long)So * tpr)), r = (int)(t - (long long)b * So * tpr), y = r / tpr, x = (r - y * tpr) * 32 + px;
        const uint8_t* L = lab + (size_t)b * S * S;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int yy = 2 * y + dy - 1;
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                const int xx = 2 * x + dx - 1;
                int l = 19;
                if ((unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S) l = L[yy * S + xx];
                l = l < 19 ? l : 19;
                const float4 v = *reinterpret_cast<const float4*>(&T[((dy * 4 + dx) * 20 + l) * 32 + cg * 4]);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        const size_t o = (((size_t)b * 32 + cg * 4) * So + y) * So + x, pcs = (size_t)So * So, pco = ((size_t)(cg * 4) * So + y) * So + x;
        out[o] = pc[pco] + acc.x;
        out[o + pcs] = pc[pco + pcs] + acc.y;
        out[o + 2 * pcs] = pc[pco + 2 * pcs] + acc.z;
        out[o + 3 * pcs] = pc[pco + 3 * pcs] + acc.w;
    }
}
hipError_t shape_enc_l0(const uint8_t* lab, const float* tab, const float* pc_hair, const float* pc_face, float* out_hair, float* out_face,
                        int B, int S, hipStream_t s) {
    if (S % 64 != 0) return hipErrorInvalidValue;
    const long long ntile = (long long)B * (S / 2) * (S / 64);
    const unsigned gx = (unsigned)(ntile < 2048 ? ntile : 2048);
    hipLaunchKernelGGL(shape_enc_l0_kernel, dim3(gx, 2), dim3(256), 0, s, lab, tab, pc_hair, pc_face, out_hair, out_face, B, S);
    return hipGetLastError();
}
hipError_t shape_inputs(const uint8_t* lab, const float* pos, float* hair_in, float* face_in, int B, int HW,
                        hipStream_t s) {
    const long long n = (long long)B * HW;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(shape_inputs_kernel, dim3(grid), dim3(256), 0, s, lab, pos, hair_in, face_in, B, HW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// nn.Linear as a batched GEMV (weight-bandwidth bound at B <= 16): out[b][o] = act(scale[o]*(bias[o] + W[o,:].x[b,:]) + shift[o])
// One wave per 2 output rows; lanes stride K (float4 when K % 4 == 0); up to LIN_BT samples per pass.
// x row stride = ldx (allows reading a slice / concatenation handled by the caller), out row stride = ldo.
constexpr int LIN_BT = 8;
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ out, int B,
                                                     int K, int O, int ldx, int ldo, int act) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o0 = (blockIdx.x * 4 + wave) * 2;
    if (o0 >= O) return;
    const bool two = o0 + 1 < O;
    const float* w0 = W + (long long)o0 * K;
    const float* w1 = W + (long long)(two ? o0 + 1 : o0) * K;
    for (int bb = 0; bb < B; bb += LIN_BT) {
        float a0[LIN_BT], a1[LIN_BT];
#pragma unroll
        for (int t = 0; t < LIN_BT; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
        if ((K & 3) == 0) {
            for (int k = lane * 4; k < K; k += 256) {
                const float4 u0 = *reinterpret_cast<const float4*>(w0 + k);
                const float4 u1 = *reinterpret_cast<const float4*>(w1 + k);
#pragma unroll
                for (int t = 0; t < LIN_BT; ++t)
                    if (bb + t < B) {
                        const float4 v = *reinterpret_cast<const float4*>(x + (long long)(bb + t) * ldx + k);
                        a0[t] += u0.x * v.x + u0.y * v.y + u0.z * v.z + u0.w * v.w;
                        a1[t] += u1.x * v.x + u1.y * v.y + u1.z * v.z + u1.w * v.w;
                    }
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float u0 = w0[k], u1 = w1[k];
#pragma unroll
                for (int t = 0; t < LIN_BT; ++t)
                    if (bb + t < B) {
                        const float v = x[(long long)(bb + t) * ldx + k];
                        a0[t] += u0 * v;
                        a1[t] += u1 * v;
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < LIN_BT; ++t) {
            a0[t] = wave_sum(a0[t]);
            a1[t] = wave_sum(a1[t]);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t)
                if (bb + t < B) {
                    float v = a0[t] + (bias ? bias[o0] : 0.f);
                    if (scale) v = v * scale[o0] + shift[o0];
                    out[(long long)(bb + t) * ldo + o0] = act_fn(v, act);
                    if (two) {
                        float u = a1[t] + (bias ? bias[o0 + 1] : 0.f);
                        if (scale) u = u * scale[o0 + 1] + shift[o0 + 1];
                        out[(long long)(bb + t) * ldo + o0 + 1] = act_fn(u, act);
                    }
                }
        }
    }
}
// One BLOCK per 2 output rows: its 4 waves split K (interleaved float4 slices when K % 4 == 0), partial sums meet in LDS and are
// added in wave order (deterministic); up to LIN_BT samples per pass.  (One wave per row pair left an 8192 -> 16 layer with 8
// waves on the whole GPU, each walking K in 32 dependent steps: 85 us for 0.5 MB of weights.)
// x row stride = ldx (allows reading a slice / concatenation handled by the caller), out row stride = ldo.
__global__ __launch_bounds__(256) void linear_ksplit_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ out, int B,
                                                     int K, int O, int ldx, int ldo, int act) {
    __shared__ float part[4][2][LIN_BT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o0 = blockIdx.x * 2;
    const bool two = o0 + 1 < O;
    const float* w0 = W + (long long)o0 * K;
    const float* w1 = W + (long long)(two ? o0 + 1 : o0) * K;
    for (int bb = 0; bb < B; bb += LIN_BT) {
        float a0[LIN_BT], a1[LIN_BT];
#pragma unroll
        for (int t = 0; t < LIN_BT; ++t) { a0[t] = 0.f; a1[t] = 0.f; }
        if ((K & 3) == 0) {
#pragma unroll 2
            for (int k = threadIdx.x * 4; k < K; k += 1024) {
                const float4 u0 = *reinterpret_cast<const float4*>(w0 + k);
                const float4 u1 = *reinterpret_cast<const float4*>(w1 + k);
#pragma unroll
                for (int t = 0; t < LIN_BT; ++t)
                    if (bb + t < B) {
                        const float4 v = *reinterpret_cast<const float4*>(x + (long long)(bb + t) * ldx + k);
                        a0[t] += u0.x * v.x + u0.y * v.y + u0.z * v.z + u0.w * v.w;
                        a1[t] += u1.x * v.x + u1.y * v.y + u1.z * v.z + u1.w * v.w;
                    }
            }
        } else {
            for (int k = threadIdx.x; k < K; k += 256) {
                const float u0 = w0[k], u1 = w1[k];
#pragma unroll
                for (int t = 0; t < LIN_BT; ++t)
                    if (bb + t < B) {
                        const float v = x[(long long)(bb + t) * ldx + k];
                        a0[t] += u0 * v;
                        a1[t] += u1 * v;
                    }
            }
        }
#pragma unroll
        for (int t = 0; t < LIN_BT; ++t) {
            a0[t] = wave_sum(a0[t]);
            a1[t] = wave_sum(a1[t]);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < LIN_BT; ++t) {
                part[wave][0][t] = a0[t];
                part[wave][1][t] = a1[t];
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * LIN_BT) {
            const int r = threadIdx.x / LIN_BT, t = threadIdx.x % LIN_BT, o = o0 + r;
            if (bb + t < B && o < O) {
                float v = ((part[0][r][t] + part[1][r][t]) + part[2][r][t]) + part[3][r][t];
                v += bias ? bias[o] : 0.f;
                if (scale) v = v * scale[o] + shift[o];
                out[(long long)(bb + t) * ldo + o] = act_fn(v, act);
            }
        }
        __syncthreads();
    }
}
hipError_t linear(const float* x, const float* W, const float* bias, const float* scale, const float* shift, float* out,
                  int B, int K, int O, int ldx, int ldo, int act, hipStream_t s) {
    // few rows and a long reduction (shape-encoder heads: 8192 -> 16 / 1024): the K-split form fills more of the chip
    if (K >= 2048 && O <= 2048)
        hipLaunchKernelGGL(linear_ksplit_kernel, dim3((O + 1) / 2), dim3(256), 0, s, x, W, bias, scale, shift, out, B, K, O, ldx, ldo, act);
    else
        hipLaunchKernelGGL(linear_kernel, dim3((O + 7) / 8), dim3(256), 0, s, x, W, bias, scale, shift, out, B, K, O, ldx, ldo, act);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Style LUT of ONE ACE at interactive batch sizes (normalization.py:117-153,172-173: conv_gamma / conv_beta of the projected codes):
//   out[n][row] = sum_k W[row][k] x[n][k],   N <= 64 (sample, label) columns, K = 512, rows = 18 C (up to 18 432).
// Weight-bandwidth bound (37.7 MB of W at C = 1024): W must stream once, at full width.  linear_kernel re-reads the N vectors from L1 for
// every pair of rows and runs at 0.6 TB/s; here the vectors sit in LDS and the sums run on the f32 matrix cores: wave = 16 rows x all N
// columns, lane (row i = lane & 15, g = lane >> 4) loads W[row][16 j + 4 g .. + 3] as ONE 16-byte load = the A operands of four MFMAs
// (k-step e of the four takes element e: the k order inside the sum is a permutation, the B operand uses the same one), the B
// operands x[n][16 j + 4 g + e] are one ds_read_b128 per 16-column tile.  Eight loads in flight per lane.
constexpr int LUTG_K = 512;
typedef float lutg_f32x4 __attribute__((ext_vector_type(4)));
constexpr int LUTG_XP = LUTG_K + 4;              // LDS row pitch of x (floats): rows 16 bytes apart in the banks
template <int NT>
__global__ __launch_bounds__(256) void lut_gemv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ W, float* __restrict__ out, int N, int rows) {
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [NT * 16][LUTG_XP]
    for (int i = threadIdx.x; i < NT * 16 * (LUTG_K / 4); i += 256) {
        const int n = i / (LUTG_K / 4), k4 = i - n * (LUTG_K / 4);
        const float4 v = n < N ? reinterpret_cast<const float4*>(x + (size_t)n * LUTG_K)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + n * LUTG_XP + 4 * k4) = v;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4;
    const int ntile = (rows + 15) / 16;
    for (int rt = blockIdx.x * 4 + wave; rt < ntile; rt += gridDim.x * 4) {
        const int row = rt * 16 + i16, rc = row < rows ? row : rows - 1;
        const float* wp = W + (size_t)rc * LUTG_K + 4 * g;
        lutg_f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (lutg_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j0 = 0; j0 < LUTG_K / 16; j0 += 8) {
            float4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = *reinterpret_cast<const float4*>(wp + 16 * (j0 + j));
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const float4 b = *reinterpret_cast<const float4*>(xs + (t * 16 + i16) * LUTG_XP + 16 * (j0 + j) + 4 * g);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b.x, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b.y, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b.z, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b.w, acc[t], 0, 0, 0);
                }
            }
        }
        // accumulator element r of lane (n = lane & 15, g): row 16 rt + 4 g + r, column n of the tile -> out[n][rows]: four consecutive rows
        const int r0 = rt * 16 + 4 * g;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = t * 16 + i16;
            if (n < N && r0 + 3 < rows) *reinterpret_cast<float4*>(out + (size_t)n * rows + r0) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            else if (n < N)
                for (int r = 0; r < 4; ++r)
                    if (r0 + r < rows) out[(size_t)n * rows + r0 + r] = acc[t][r];
        }
    }
}
hipError_t lut_gemv_mfma(const float* x, const float* W, float* out, int N, int rows, hipStream_t s) {
    if (N < 1 || N > 64 || rows < 1) return hipErrorInvalidValue;
    const int nt = (N + 15) / 16;
    const size_t lds = (size_t)nt * 16 * LUTG_XP * sizeof(float);
    const int ntile = (rows + 15) / 16;
    const int grid = (ntile + 3) / 4 < 1024 ? (ntile + 3) / 4 : 1024;
    static bool attr_done[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
#define LUTG_LAUNCH(NT_)                                                                                                                          \
    {                                                                                                                                             \
        if (!attr_done[NT_ - 1][dev]) {                                                                                                           \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lut_gemv_mfma_kernel<NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)(NT_ * 16 * LUTG_XP * sizeof(float)));                                                        \
            if (e != hipSuccess) return e;                                                                                                        \
            attr_done[NT_ - 1][dev] = true;                                                                                                       \
        }                                                                                                                                         \
        hipLaunchKernelGGL(lut_gemv_mfma_kernel<NT_>, dim3(grid), dim3(256), lds, s, x, W, out, N, rows);                                         \
    }
    if (nt == 1) LUTG_LAUNCH(1)
    else if (nt == 2) LUTG_LAUNCH(2)
    else if (nt == 3) LUTG_LAUNCH(3)
    else LUTG_LAUNCH(4)
#undef LUTG_LAUNCH
    return hipGetLastError();
}

// EigenGAN subspace injection (model_eigengan.py:27-31,76-81): h[b,:] = lrelu(h[b,:] + U (L * z[b,:]) + mu), z 2-d
__global__ void subspace_add_kernel(float* __restrict__ h, const float* __restrict__ z, int zld, const float* __restrict__ U,
                                    const float* __restrict__ L, const float* __restrict__ mu, int B, int D, int Z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i % D;
    float v = h[i] + mu[d];
    for (int k = 0; k < Z; ++k) v += U[k * D + d] * L[k] * z[b * zld + k];   // U: [Z][D]
    h[i] = v > 0.f ? v : 0.2f * v;
}
hipError_t subspace_add(float* h, const float* z, int zld, const float* U, const float* L, const float* mu, int B, int D,
                        int Z, hipStream_t s) {
    hipLaunchKernelGGL(subspace_add_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, h, z, zld, U, L, mu, B, D, Z);
    return hipGetLastError();
}

}  // namespace chk
