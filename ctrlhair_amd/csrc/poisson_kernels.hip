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
// ~1e-2 relative error, more than one uint8 level after the gamma power).  L2-resident streaming kernels, 2 launches per iteration
// (Chronopoulos-Gear CG), scalars (alpha, beta, convergence) stay on the device.  Dot products are two-level and
// deterministic: per-block partial sums, re-reduced in a fixed order by every block that needs the scalar (no atomics).
#include <hip/hip_runtime.h>

#include <climits>
#include <stdint.h>

#include "kernels.h"

namespace chk {

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

// gamma transform, right-hand side, initial guess x0 = target, r0 = b - A x0, p = s = 0
__global__ __launch_bounds__(256) void pb_setup_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ tgt,
                                                       const uint8_t* __restrict__ mask, double* __restrict__ X,
                                                       double* __restrict__ R, double* __restrict__ P,
                                                       double* __restrict__ S, uint8_t* __restrict__ U, int H, int W,
                                                       float inv_gamma) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (k < HW) {
        const int y = k / W, x = k % W;
        const bool unk = pb_unknown(mask, y, x, H, W);
        U[k] = unk ? 1 : 0;
        const int ny[4] = {y, y, y + 1, y - 1}, nx[4] = {x + 1, x - 1, x, x};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double t = pow((double)tgt[k * 3 + c], (double)inv_gamma);
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
            P[c * HW + k] = 0.0;
            S[c * HW + k] = 0.0;
        }
    }
}

// ---- CG in the Chronopoulos-Gear form: one matvec and ONE fused pair of dot products per iteration, hence two kernel
// launches per iteration (the solver is launch-bound: the vectors of a 512x512 image live in L2):
//      w = A r,  gamma = r.r,  delta = r.w                                        (pb_matvec_kernel)
//      beta = gamma / gamma_prev,  alpha = gamma / (delta - beta gamma / alpha_prev)   [beta = 0, alpha = gamma/delta at i = 0]
//      p = r + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s              (pb_update_kernel)
// Every block re-reduces the per-block partials in the same order, so all blocks agree bit for bit on the scalars and on
// convergence (no flag race); the previous iteration's scalars sit in a ping-pong slot written by block 0.
constexpr int PB_MAXBLK = 512;      // CG kernels: grid-stride over at most this many blocks (2 per CU)

struct PoissonCG {
    double gamma[2][3], alpha[2][3], gamma0[3];
    int done, iters;
};

__global__ void pb_init_cg_kernel(PoissonCG* cg) {
    for (int q = 0; q < 2; ++q)
        for (int c = 0; c < 3; ++c) cg->gamma[q][c] = cg->alpha[q][c] = 0.0;
    for (int c = 0; c < 3; ++c) cg->gamma0[c] = 0.0;
    cg->done = 0;
    cg->iters = 0;
}

// w = A r on the unknowns; partG[c][block] = partial r.r, partD[c][block] = partial r.w
__global__ __launch_bounds__(256) void pb_matvec_kernel(const double* __restrict__ R, double* __restrict__ Wv,
                                                        const uint8_t* __restrict__ U, const PoissonCG* cg,
                                                        double* __restrict__ partG, double* __restrict__ partD, int H, int W) {
    if (cg->done) return;
    __shared__ double sh[4];
    const int HW = H * W;
    double g[3] = {0.0, 0.0, 0.0}, d[3] = {0.0, 0.0, 0.0};
    // grid-stride: at most PB_MAXBLK blocks, so that the per-block partials every block re-reduces stay a few KiB
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < HW; k += gridDim.x * blockDim.x) {
        if (!U[k]) continue;
        const int y = k / W, x = k % W;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double* r = R + c * HW;
            double v = 4.0 * r[k];
            if (x + 1 < W && U[k + 1]) v -= r[k + 1];
            if (x > 0 && U[k - 1]) v -= r[k - 1];
            if (y + 1 < H && U[k + W]) v -= r[k + W];
            if (y > 0 && U[k - W]) v -= r[k - W];
            Wv[c * HW + k] = v;
            g[c] += r[k] * r[k];
            d[c] += r[k] * v;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double tg = pb_block_sum(g[c], sh), td = pb_block_sum(d[c], sh);
        if (threadIdx.x == 0) {
            partG[c * gridDim.x + blockIdx.x] = tg;
            partD[c * gridDim.x + blockIdx.x] = td;
        }
    }
}

__global__ __launch_bounds__(256) void pb_update_kernel(double* __restrict__ X, double* __restrict__ R, double* __restrict__ P,
                                                        double* __restrict__ S, const double* __restrict__ Wv,
                                                        const uint8_t* __restrict__ U, PoissonCG* cg,
                                                        const double* __restrict__ partG, const double* __restrict__ partD,
                                                        int HW, int iter, double rel_tol2) {
    if (cg->done) return;
    __shared__ double sh[4];
    const int q = iter & 1;
    double alpha[3], beta[3], gam[3];
    bool live[3], any = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gam[c] = pb_total(partG + c * gridDim.x, gridDim.x, sh);
        const double del = pb_total(partD + c * gridDim.x, gridDim.x, sh);
        const double g0 = iter == 0 ? gam[c] : cg->gamma0[c];
        live[c] = gam[c] > rel_tol2 * g0 && gam[c] > 1e-24;      // channel not yet converged
        if (iter == 0) {
            beta[c] = 0.0;
            alpha[c] = del > 0.0 ? gam[c] / del : 0.0;
        } else {
            const double gp = cg->gamma[q ^ 1][c], ap = cg->alpha[q ^ 1][c];
            beta[c] = gp > 0.0 ? gam[c] / gp : 0.0;
            const double den = del - (ap != 0.0 ? beta[c] * gam[c] / ap : 0.0);
            alpha[c] = den > 0.0 ? gam[c] / den : 0.0;
        }
        if (!live[c]) alpha[c] = beta[c] = 0.0;
        any = any || live[c];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {                     // bookkeeping for the next iteration / the host
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            cg->gamma[q][c] = live[c] ? gam[c] : cg->gamma[q ^ 1][c];
            cg->alpha[q][c] = live[c] ? alpha[c] : cg->alpha[q ^ 1][c];
            if (iter == 0) cg->gamma0[c] = gam[c];
        }
        if (any) cg->iters = iter + 1;
        else cg->done = 1;
    }
    if (!any) return;                                              // same decision in every block
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < HW; k += gridDim.x * blockDim.x) {
        if (!U[k]) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (!live[c]) continue;                                // a converged channel keeps its solution
            const int i = c * HW + k;
            const double p = R[i] + beta[c] * P[i];
            const double s = Wv[i] + beta[c] * S[i];
            P[i] = p;
            S[i] = s;
            X[i] += alpha[c] * p;
            R[i] -= alpha[c] * s;
        }
    }
}

// Residual check after the LAST allowed update (the update kernel only learns at the top of the following iteration that the
// previous one met the tolerance): sums the partials of a final mat-vec and sets the flag.  One block.  nparts = blocks of
// that mat-vec.  With no iteration done (max_iters = 0) there is no ||r0|| on file: only an exactly solved system counts.
__global__ __launch_bounds__(256) void pb_final_check_kernel(PoissonCG* __restrict__ cg, const double* __restrict__ partG, int nparts,
                                                             double rel_tol2) {
    __shared__ double sh[4];
    bool any = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double gam = pb_total(partG + c * nparts, nparts, sh);
        const double g0 = cg->gamma0[c];
        any = any || (gam > rel_tol2 * g0 && gam > 1e-24);
    }
    if (threadIdx.x == 0 && !any) cg->done = 1;
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
    return 5 * 3 * HW * sizeof(double) + 2 * 3 * PB_MAXBLK * sizeof(double) + HW + 512 + sizeof(PoissonCG);
}

hipError_t poisson_blend(const uint8_t* src, const uint8_t* tgt, const uint8_t* mask, uint8_t* out, int H, int W,
                         int with_gamma, int max_iters, double rel_tol, void* ws, int* iters_out, hipStream_t s) {
    const int HW = H * W;
    double* X = static_cast<double*>(ws);
    double *R = X + 3 * (size_t)HW, *P = R + 3 * (size_t)HW, *S = P + 3 * (size_t)HW, *Wv = S + 3 * (size_t)HW;
    const int nb = (HW + 255) / 256, ncg = nb < PB_MAXBLK ? nb : PB_MAXBLK;
    double *partG = Wv + 3 * (size_t)HW, *partD = partG + 3 * (size_t)PB_MAXBLK;
    uint8_t* U = reinterpret_cast<uint8_t*>(partD + 3 * (size_t)PB_MAXBLK);
    PoissonCG* cg = reinterpret_cast<PoissonCG*>(U + (((size_t)HW + 255) / 256) * 256);
    const float gamma = with_gamma ? 2.2f : 1.0f;
    const dim3 g(nb), gc(ncg), b(256);
    hipLaunchKernelGGL(pb_init_cg_kernel, dim3(1), dim3(1), 0, s, cg);
    hipLaunchKernelGGL(pb_setup_kernel, g, b, 0, s, src, tgt, mask, X, R, P, S, U, H, W, 1.0f / gamma);
    int done = 0;
    for (int it = 0; it < max_iters && !done; ++it) {
        hipLaunchKernelGGL(pb_matvec_kernel, gc, b, 0, s, R, Wv, U, cg, partG, partD, H, W);
        hipLaunchKernelGGL(pb_update_kernel, gc, b, 0, s, X, R, P, S, Wv, U, cg, partG, partD, HW, it, rel_tol * rel_tol);
        if ((it & 63) == 63) {        // poll the device flag every 64 iterations (converged runs stop launching)
            hipError_t e = hipMemcpyAsync(&done, &cg->done, sizeof(int), hipMemcpyDeviceToHost, s);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(s);
            if (e != hipSuccess) return e;
        }
    }
    if (!done) {           // did the last allowed update reach the tolerance?  (mat-vec + reduction only, no update)
        hipLaunchKernelGGL(pb_matvec_kernel, gc, b, 0, s, R, Wv, U, cg, partG, partD, H, W);
        hipLaunchKernelGGL(pb_final_check_kernel, dim3(1), b, 0, s, cg, partG, ncg, rel_tol * rel_tol);
    }
    hipLaunchKernelGGL(pb_finish_kernel, g, b, 0, s, X, out, HW, gamma);
    if (iters_out) {       // iteration count; negated (INT_MIN for zero iterations) when rel_tol was not reached
        int dn = 0, its = 0;
        hipError_t e = hipMemcpyAsync(&dn, &cg->done, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&its, &cg->iters, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        *iters_out = dn ? its : (its > 0 ? -its : INT_MIN);
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
