// tools/interior_bench.hip -- stand-alone timing of the f16x3 interior pass (ace_interior_sh16_kernel) against a byte-for-byte
// copy kernel of the same access pattern.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ctrlhair_amd/csrc
//   tools/interior_bench.hip -o tools/interior_bench.bin ; run on the GPU box.
#include "../ctrlhair_amd/csrc/ace_sparse.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>

using namespace chk;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// same bytes, no table / math: one thread = one pixel, 8 channels per step
__global__ __launch_bounds__(256) void copy_like_kernel(const float4* __restrict__ x, uint4* __restrict__ out, const uint8_t* __restrict__ u5,
                                                        int B, int C, int H, int W, int x_up) {
    const int HW = H * W, ppb = (HW + 255) / 256, nblk = B * ppb;
    const int xW = W >> x_up, xHW = xW * (H >> x_up), Go = C >> 3;
    for (int pbk = blockIdx.x; pbk < nblk; pbk += gridDim.x) {
        const int b = pbk / ppb, pix = (pbk - b * ppb) * 256 + threadIdx.x;
        const int y = pix / W, xx = pix - y * W;
        if (u5[(long long)b * HW + pix] >= 19) continue;
        const float4* xp = x + (long long)b * (C >> 2) * xHW + (long long)(y >> x_up) * xW + (xx >> x_up);
        uint4* op = out + (long long)b * Go * 2 * HW + pix;
        for (int g = 0; g < Go; ++g) {
            const float4 a = xp[(long long)(2 * g) * xHW], d = xp[(long long)(2 * g + 1) * xHW];
            uint4 h, l;
            h.x = __float_as_uint(a.x); h.y = __float_as_uint(a.y); h.z = __float_as_uint(a.z); h.w = __float_as_uint(a.w);
            l.x = __float_as_uint(d.x); l.y = __float_as_uint(d.y); l.z = __float_as_uint(d.z); l.w = __float_as_uint(d.w);
            op[(long long)g * 2 * HW] = h;
            op[(long long)g * 2 * HW + HW] = l;
        }
    }
}

int main(int argc, char** argv) {
    struct Cfg { int B, C, H, up; };
    const Cfg cfgs[] = {{16, 64, 512, 1}, {16, 64, 512, 0}, {16, 128, 256, 1}, {16, 128, 256, 0}, {16, 256, 128, 1}};
    for (int stripes = 0; stripes < 3; ++stripes)
    for (const Cfg& c : cfgs) {
        const int B = c.B, C = c.C, H = c.H, W = c.H, HW = H * W, xh = H >> c.up;
        std::vector<uint8_t> u5((size_t)B * HW);
        size_t ninter = 0;
        for (int b = 0; b < B; ++b)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int cell = stripes == 2 ? H / 16 : H / 8;   // blocky labels: 8 x 8 (16 x 16) cells, 4-pixel boundary bands
                    uint8_t v = (uint8_t)(((x / cell) + (y / cell) * 3 + b) % 19);
                    if (stripes && ((x % cell) < 2 || (x % cell) >= cell - 2 || (y % cell) < 2 || (y % cell) >= cell - 2)) v = 255;
                    u5[((size_t)b * H + y) * W + x] = v;
                    ninter += v < 19;
                }
        uint8_t* d_u5; float *d_x, *d_gtab, *d_par, *d_noise; void* d_out; int* d_cnt; unsigned* d_amax;
        CK(hipMalloc(&d_u5, u5.size()));
        CK(hipMemcpy(d_u5, u5.data(), u5.size(), hipMemcpyHostToDevice));
        const size_t nx = (size_t)B * C * xh * xh, nout = (size_t)B * C * HW;
        CK(hipMalloc(&d_x, nx * 4)); CK(hipMemset(d_x, 0, nx * 4));
        CK(hipMalloc(&d_out, nout * 4));
        CK(hipMalloc(&d_gtab, (size_t)B * 19 * 2 * C * 4)); CK(hipMemset(d_gtab, 0, (size_t)B * 19 * 2 * C * 4));
        CK(hipMalloc(&d_par, 3 * C * 4)); CK(hipMemset(d_par, 0, 3 * C * 4));
        CK(hipMalloc(&d_noise, (size_t)B * HW * 4)); CK(hipMemset(d_noise, 0, (size_t)B * HW * 4));
        CK(hipMalloc(&d_cnt, 4096 * 16 * 4)); CK(hipMemset(d_cnt, 0, 4096 * 16 * 4));
        CK(hipMalloc(&d_amax, 8)); CK(hipMemset(d_amax, 0, 8));
        AceInteriorParams q{};
        q.x = d_x; q.out = d_out; q.u5 = d_u5; q.gtab = d_gtab; q.bn_a = d_par; q.bn_d = d_par + C; q.nv = d_par + 2 * C;
        q.noise = d_noise; q.noise_bstride = HW; q.B = B; q.C = C; q.H = H; q.W = W; q.x_up = c.up; q.act = 1;
        q.out_scale = 1.f; q.out_amax = d_amax; q.pass = 0; q.bf16 = 0; q.cnt = d_cnt; q.variant = 1;
        {   // random operands
            std::vector<float> h(nx);
            for (auto& v : h) v = (float)rand() / RAND_MAX * 4.f - 2.f;
            CK(hipMemcpy(d_x, h.data(), nx * 4, hipMemcpyHostToDevice));
            h.resize((size_t)B * 19 * 2 * C);
            for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
            CK(hipMemcpy(d_gtab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            h.resize(3 * C);
            for (auto& v : h) v = (float)rand() / RAND_MAX + 0.5f;
            CK(hipMemcpy(d_par, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            h.resize((size_t)B * HW);
            for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
            CK(hipMemcpy(d_noise, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        // correctness: the tile kernel (masked and filling) must write the interior pixels exactly like the row kernel
        std::vector<uint32_t> ref(nout), got(nout);
        CK(hipMemset(d_out, 0, nout * 4));
        q.impl = 1; CK(ace_interior_sh16(q, 0)); CK(hipMemcpy(ref.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
        unsigned amax_ref = 0, amax_got = 0;
        CK(hipMemcpy(&amax_ref, d_amax, 4, hipMemcpyDeviceToHost));
        for (int fm : {257, 1}) {
            CK(hipMemset(d_out, 0, nout * 4)); CK(hipMemset(d_amax, 0, 8));
            q.impl = 0; q.fill_min = fm; CK(ace_interior_sh16(q, 0)); CK(hipMemcpy(got.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&amax_got, d_amax, 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (int b = 0; b < B; ++b)
                for (int u = 0; u < C / 8 * 2; ++u)
                    for (int p = 0; p < HW; ++p) {
                        if (u5[(size_t)b * HW + p] >= 19) continue;
                        const size_t o = (((size_t)b * (C / 8 * 2) + u) * HW + p) * 4;
                        for (int k = 0; k < 4; ++k) bad += ref[o + k] != got[o + k];
                    }
            printf("  check fill_min=%3d: %zu mismatching words on interior pixels, amax %08x vs %08x\n", fm, bad, amax_got, amax_ref);
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const double bytes = (double)ninter * C * (4.0 / (c.up ? 4 : 1) + 4.0);
        const char* names[] = {"rows (old)   ", "tiles masked ", "tiles fill192", "tiles fill128", "tiles fill 64", "copy-like    "};
        for (int which = 0; which < 6; ++which) {
            float best = 1e9f;
            q.impl = which == 0 ? 1 : 0;
            q.fill_min = which == 1 ? 257 : (which == 2 ? 192 : (which == 3 ? 128 : 64));
            for (int it = 0; it < 6; ++it) {
                CK(hipEventRecord(e0, 0));
                if (which < 5) CK(ace_interior_sh16(q, 0));
                else hipLaunchKernelGGL(copy_like_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)d_x, (uint4*)d_out, d_u5, B, C, H, W, c.up);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (it > 0 && ms < best) best = ms;
            }
            printf("%s stripes=%d B=%d C=%3d H=%3d up=%d interior=%.2f : %8.1f us  %7.1f GB/s (algorithmic bytes of the interior pixels)\n",
                   names[which], stripes, B, C, H, c.up, (double)ninter / ((double)B * HW), best * 1e3, bytes / best * 1e-6);
        }
        {   // exact-f32 kernels on the same operands (x read as NCHW: same byte count)
            std::vector<uint32_t> r32(nout), g32(nout);
            CK(hipMemset(d_out, 0, nout * 4));
            q.impl = 1; q.variant = 0; CK(ace_interior_f32(q, 0)); CK(hipMemcpy(r32.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
            for (int fm : {257, 1}) {
                CK(hipMemset(d_out, 0, nout * 4));
                q.impl = 0; q.fill_min = fm; CK(ace_interior_f32(q, 0)); CK(hipMemcpy(g32.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (int b = 0; b < B; ++b)
                    for (int ch = 0; ch < C; ++ch)
                        for (int p = 0; p < HW; ++p)
                            if (u5[(size_t)b * HW + p] < 19) bad += r32[((size_t)b * C + ch) * HW + p] != g32[((size_t)b * C + ch) * HW + p];
                printf("  f32 check fill_min=%3d: %zu mismatching words on interior pixels\n", fm, bad);
            }
            for (int fm : {257, 1}) {
                CK(hipMemset(d_out, 0, nout * 4));
                q.impl = 2; q.fill_min = fm; CK(ace_interior_f32(q, 0)); CK(hipMemcpy(g32.data(), d_out, nout * 4, hipMemcpyDeviceToHost));
                size_t bad = 0;
                for (int b = 0; b < B; ++b)
                    for (int ch = 0; ch < C; ++ch)
                        for (int p = 0; p < HW; ++p)
                            if (u5[(size_t)b * HW + p] < 19) bad += r32[((size_t)b * C + ch) * HW + p] != g32[((size_t)b * C + ch) * HW + p];
                printf("  f32 tile4 check fill_min=%3d: %zu mismatching words on interior pixels\n", fm, bad);
            }
            const char* n32[] = {"f32 rows (old)   ", "f32 tiles masked ", "f32 tiles fill128", "f32 tile4 masked ", "f32 tile4 fill128", "f32 tile4 fill 64"};
            for (int which = 0; which < 6; ++which) {
                float best = 1e9f;
                q.impl = which == 0 ? 1 : (which < 3 ? 0 : 2);
                q.fill_min = (which == 1 || which == 3) ? 257 : (which == 5 ? 64 : 128);
                for (int it = 0; it < 6; ++it) {
                    CK(hipEventRecord(e0, 0));
                    CK(ace_interior_f32(q, 0));
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (it > 0 && ms < best) best = ms;
                }
                printf("%s stripes=%d B=%d C=%3d H=%3d up=%d interior=%.2f : %8.1f us  %7.1f GB/s\n",
                       n32[which], stripes, B, C, H, c.up, (double)ninter / ((double)B * HW), best * 1e3, bytes / best * 1e-6);
            }
            q.variant = 1;
        }
        hipFree(d_u5); hipFree(d_x); hipFree(d_out); hipFree(d_gtab); hipFree(d_par); hipFree(d_noise); hipFree(d_cnt); hipFree(d_amax);
    }
    return 0;
}
