// poisson_kernels.hip -- the blending step that follows the generator (hair_editor.py:285-310, poisson_blending.py:29-87)
// on the GPU: the reference assembles a (HW)^2 sparse matrix with a Python double loop and calls a sparse direct solver
// three times (seconds per image); here the same linear system is solved matrix-free by conjugate gradients.
//
// System (per colour channel, gamma space v = u8^(1/2.2)):  unknowns U = {mask != 0} + {border pixels} (the reference leaves
// border pixels outside the mask as Laplacian rows with the target value as right-hand side, poisson_blending.py:48-56);
// known K = interior pixels with mask == 0, x = target.  For k in U:
//      4 x_k - sum_{n in N4(k), in image, n in U} x_n  =  base_k + sum_{n in N4(k), n in K} target_n
//      base_k = (L source)_k  if mask_k != 0  else target_k,      (L s)_k = 4 s_k - sum_{n in N4(k), in image} s_n
// Symmetric positive definite -> CG; all vectors in f64 (condition number ~ (2N/pi)^2 ~ 1e5 at 512 px: an f32 CG stalls at
// ~1e-2 relative error, more than one uint8 level after the gamma power).  HBM/L2-bound streaming kernels, 3 launches per
// iteration, scalars (alpha, beta, convergence flag) stay on the device.  Dot products are two-level and
// deterministic: per-block partial sums, re-reduced in a fixed order by every block that needs the scalar (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace chk {

struct PoissonScalars {
    double rs_old[3], rs0[3];
    int done, iters;
};

// sum of v over the 256-thread block, returned to every thread (fixed order -> run-to-run deterministic)
__device__ __forceinline__ double pb_block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// total of the per-block partials part[0 .. n) (every block computes the same value in the same order)
__device__ __forceinline__ double pb_total(const double* __restrict__ part, int n, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += part[i];
    return pb_block_sum(v, sh);
}

__device__ __forceinline__ bool pb_unknown(const uint8_t* m, int y, int x, int H, int W) {
    return m[y * W + x] != 0 || y == 0 || x == 0 || y == H - 1 || x == W - 1;
}

// gamma transform, right-hand side, initial guess x0 = target, r0 = p0 = b - A x0, rs_old = r0.r0
__global__ __launch_bounds__(256) void pb_setup_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ tgt,
                                                       const uint8_t* __restrict__ mask, double* __restrict__ X,
                                                       double* __restrict__ R, double* __restrict__ P,
                                                       double* __restrict__ T, uint8_t* __restrict__ U,
                                                       double* __restrict__ partB, int H, int W, float inv_gamma) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    double acc[3] = {0.0, 0.0, 0.0};
    if (k < HW) {
        const int y = k / W, x = k % W;
        const bool unk = pb_unknown(mask, y, x, H, W);
        U[k] = unk ? 1 : 0;
        const int ny[4] = {y, y, y + 1, y - 1}, nx[4] = {x + 1, x - 1, x, x};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double t = pow((double)tgt[k * 3 + c], (double)inv_gamma);
            T[c * HW + k] = t;
            X[c * HW + k] = t;
            double r = 0.0;
            if (unk) {
                // b_k
                double b;
                if (mask[k] != 0) {
                    b = 4.0 * pow((double)src[k * 3 + c], (double)inv_gamma);
                    for (int q = 0; q < 4; ++q)
                        if ((unsigned)ny[q] < (unsigned)H && (unsigned)nx[q] < (unsigned)W)
                            b -= pow((double)src[(ny[q] * W + nx[q]) * 3 + c], (double)inv_gamma);
                } else {
                    b = t;
                }
                // r0 = b + sum_{known nbrs} t_n - (4 t_k - sum_{unknown nbrs} t_n) = b - 4 t_k + sum_{all in-image nbrs} t_n
                r = b - 4.0 * t;
                for (int q = 0; q < 4; ++q)
                    if ((unsigned)ny[q] < (unsigned)H && (unsigned)nx[q] < (unsigned)W)
                        r += pow((double)tgt[(ny[q] * W + nx[q]) * 3 + c], (double)inv_gamma);
            }
            R[c * HW + k] = r;
            P[c * HW + k] = r;
            acc[c] = r * r;
        }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t = pb_block_sum(acc[c], sh);
        if (threadIdx.x == 0) partB[c * gridDim.x + blockIdx.x] = t;
    }
}

__global__ void pb_init_scalars_kernel(PoissonScalars* sc) {
    for (int c = 0; c < 3; ++c) sc->rs_old[c] = sc->rs0[c] = 0.0;
    sc->done = 0;
    sc->iters = 0;
}

// Ap = A p on the unknowns; partA[c][block] = partial p.Ap
__global__ __launch_bounds__(256) void pb_matvec_kernel(const double* __restrict__ P, double* __restrict__ AP,
                                                        const uint8_t* __restrict__ U, const PoissonScalars* sc,
                                                        double* __restrict__ partA, int H, int W) {
    if (sc->done) return;
    __shared__ double sh[4];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    double acc[3] = {0.0, 0.0, 0.0};
    if (k < HW && U[k]) {
        const int y = k / W, x = k % W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double* p = P + c * HW;
            double v = 4.0 * p[k];
            if (x + 1 < W && U[k + 1]) v -= p[k + 1];
            if (x > 0 && U[k - 1]) v -= p[k - 1];
            if (y + 1 < H && U[k + W]) v -= p[k + W];
            if (y > 0 && U[k - W]) v -= p[k - W];
            AP[c * HW + k] = v;
            acc[c] = v * p[k];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t = pb_block_sum(acc[c], sh);
        if (threadIdx.x == 0) partA[c * gridDim.x + blockIdx.x] = t;
    }
}

// x += alpha p; r -= alpha Ap (alpha = rs_old / sum(partA)); partB[c][block] = partial r.r
__global__ __launch_bounds__(256) void pb_update_kernel(double* __restrict__ X, double* __restrict__ R,
                                                        const double* __restrict__ P, const double* __restrict__ AP,
                                                        const uint8_t* __restrict__ U, const PoissonScalars* sc,
                                                        const double* __restrict__ partA, double* __restrict__ partB, int HW) {
    if (sc->done) return;
    __shared__ double sh[4];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double alpha[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double pap = pb_total(partA + c * gridDim.x, gridDim.x, sh);
        alpha[c] = pap > 0.0 ? sc->rs_old[c] / pap : 0.0;
    }
    double acc[3] = {0.0, 0.0, 0.0};
    if (k < HW && U[k]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X[c * HW + k] += alpha[c] * P[c * HW + k];
            const double r = R[c * HW + k] - alpha[c] * AP[c * HW + k];
            R[c * HW + k] = r;
            acc[c] = r * r;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t = pb_block_sum(acc[c], sh);
        if (threadIdx.x == 0) partB[c * gridDim.x + blockIdx.x] = t;
    }
}

// p = r + beta p (beta = sum(partB) / rs_old)
__global__ __launch_bounds__(256) void pb_direction_kernel(double* __restrict__ P, const double* __restrict__ R,
                                                           const uint8_t* __restrict__ U, const PoissonScalars* sc,
                                                           const double* __restrict__ partB, int HW) {
    if (sc->done) return;
    __shared__ double sh[4];
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double beta[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double rs_new = pb_total(partB + c * gridDim.x, gridDim.x, sh);
        beta[c] = sc->rs_old[c] > 0.0 ? rs_new / sc->rs_old[c] : 0.0;
    }
    if (k < HW && U[k]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) P[c * HW + k] = R[c * HW + k] + beta[c] * P[c * HW + k];
    }
}

// one block, between iterations: rs_old <- sum(partB) (= r.r after the update), convergence test ||r||^2 <= tol^2 ||r0||^2
__global__ __launch_bounds__(256) void pb_roll_kernel(PoissonScalars* sc, const double* __restrict__ partB, int nblocks,
                                                      double rel_tol2, int first) {
    if (sc->done) return;
    __shared__ double sh[4];
    double rs[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rs[c] = pb_total(partB + c * nblocks, nblocks, sh);
    if (threadIdx.x != 0) return;
    bool conv = true;
    for (int c = 0; c < 3; ++c) {
        sc->rs_old[c] = rs[c];
        if (first) sc->rs0[c] = rs[c];
        if (rs[c] > rel_tol2 * sc->rs0[c] && rs[c] > 1e-24) conv = false;
    }
    if (!first) sc->iters += 1;
    if (conv) sc->done = 1;
}

// out = clamp(x^gamma) truncated to uint8 (poisson_blending.py:81-86); NaN (negative base) -> 0
__global__ void pb_finish_kernel(const double* __restrict__ X, uint8_t* __restrict__ out, int HW, float gamma) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= HW) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double x = X[c * HW + k];
        double v = x < 0.0 ? 0.0 : pow(x, (double)gamma);
        v = v > 255.0 ? 255.0 : v;
        out[k * 3 + c] = (uint8_t)v;
    }
}

size_t poisson_workspace_bytes(int H, int W) {
    const size_t HW = (size_t)H * W;
    const size_t nb = (HW + 255) / 256;
    return 5 * 3 * HW * sizeof(double) + 2 * 3 * nb * sizeof(double) + HW + 512 + sizeof(PoissonScalars);
}

hipError_t poisson_blend(const uint8_t* src, const uint8_t* tgt, const uint8_t* mask, uint8_t* out, int H, int W,
                         int with_gamma, int max_iters, double rel_tol, void* ws, int* iters_out, hipStream_t s) {
    const int HW = H * W;
    double* X = static_cast<double*>(ws);
    double *R = X + 3 * (size_t)HW, *P = R + 3 * (size_t)HW, *AP = P + 3 * (size_t)HW, *T = AP + 3 * (size_t)HW;
    const int nb = (HW + 255) / 256;
    double *partA = T + 3 * (size_t)HW, *partB = partA + 3 * (size_t)nb;
    uint8_t* U = reinterpret_cast<uint8_t*>(partB + 3 * (size_t)nb);
    PoissonScalars* sc = reinterpret_cast<PoissonScalars*>(U + (((size_t)HW + 255) / 256) * 256);
    const float gamma = with_gamma ? 2.2f : 1.0f;
    const dim3 g(nb), b(256);
    hipLaunchKernelGGL(pb_init_scalars_kernel, dim3(1), dim3(1), 0, s, sc);
    hipLaunchKernelGGL(pb_setup_kernel, g, b, 0, s, src, tgt, mask, X, R, P, T, U, partB, H, W, 1.0f / gamma);
    hipLaunchKernelGGL(pb_roll_kernel, dim3(1), b, 0, s, sc, partB, nb, rel_tol * rel_tol, 1);
    int done = 0;
    for (int it = 0; it < max_iters && !done; ++it) {
        hipLaunchKernelGGL(pb_matvec_kernel, g, b, 0, s, P, AP, U, sc, partA, H, W);
        hipLaunchKernelGGL(pb_update_kernel, g, b, 0, s, X, R, P, AP, U, sc, partA, partB, HW);
        hipLaunchKernelGGL(pb_direction_kernel, g, b, 0, s, P, R, U, sc, partB, HW);
        hipLaunchKernelGGL(pb_roll_kernel, dim3(1), b, 0, s, sc, partB, nb, rel_tol * rel_tol, 0);
        if ((it & 63) == 63) {        // poll the device flag every 64 iterations (converged runs stop launching)
            hipError_t e = hipMemcpyAsync(&done, &sc->done, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
        }
    }
    hipLaunchKernelGGL(pb_finish_kernel, g, b, 0, s, X, out, HW, gamma);
    if (iters_out) {
        hipError_t e = hipMemcpyAsync(iters_out, &sc->iters, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// ---- blending mask (hair_editor.py:297-305) ------------------------------------------------------------------------
// res = dilate13(hair) outside the target's background, dilate5(hair) on it; hair = (target == 13) | (face == 13).
// Structuring elements: OpenCV MORPH_ELLIPSE rows, half-widths per |dy| passed in constant tables.
__constant__ int c_hw13[13] = {0, 3, 4, 5, 6, 6, 6, 6, 6, 5, 4, 3, 0};
__constant__ int c_hw5[5] = {0, 2, 2, 2, 0};

__global__ void blend_mask_kernel(const uint8_t* __restrict__ tp, const uint8_t* __restrict__ fp, uint8_t* __restrict__ out,
                                  int H, int W) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= H * W) return;
    const int y = k / W, x = k % W;
    const bool bg = tp[k] == 0;
    const int r = bg ? 2 : 6;
    uint8_t v = 0;
    for (int dy = -r; dy <= r && !v; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        const int hw = bg ? c_hw5[dy + 2] : c_hw13[dy + 6];
        for (int dx = -hw; dx <= hw; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const int q = yy * W + xx;
            if (tp[q] == 13 || fp[q] == 13) { v = 1; break; }
        }
    }
    out[k] = v;
}

hipError_t blend_mask(const uint8_t* target_parsing, const uint8_t* face_parsing, uint8_t* out, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(blend_mask_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, target_parsing, face_parsing, out, H, W);
    return hipGetLastError();
}

}  // namespace chk
