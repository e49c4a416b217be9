// conv_inst_wino4.hip -- instantiation + launcher of the Winograd F(4x4,3x3) exact-f32 MFMA kernel (conv_wino4.h)
#include "conv_wino4v.h"

namespace chk {

hipError_t conv_wino4_plain(Wino4Params p, hipStream_t s) {
    if (!wino4_supported(p.H, p.W, p.Cin) || !p.in || !p.wpk || !p.out || (p.reflect && (p.res_up || p.H < 2 || p.W < 2))) return hipErrorInvalidValue;
    wino4_fill_launch(p);
    static bool done[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_plain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_plain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4::LDS_BYTES);
        if (e != hipSuccess) return e;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
        done[dev] = true;
    }
    const int grid = p.ntasks < cus[dev] ? p.ntasks : cus[dev];
    if (p.reflect) hipLaunchKernelGGL(wino4_plain_kernel<1>, dim3(grid), dim3(512), wino4::LDS_BYTES, s, p);
    else hipLaunchKernelGGL(wino4_plain_kernel<0>, dim3(grid), dim3(512), wino4::LDS_BYTES, s, p);
    return hipGetLastError();
}

hipError_t conv_wino4_ace(Wino4AceParams p, hipStream_t s) {
    if (!wino4_ace_supported(p.H, p.W, p.C) || !p.actv || !p.wpk || !p.out || !p.x || !p.noise) return hipErrorInvalidValue;
    p.nrt = (p.C + 15) / 16;
    p.ntx = p.W / wino4::TS;
    p.nty = p.H / wino4::TS;
    p.ntiles = p.B * p.ntx * p.nty;
    p.ntasks = p.ntiles * p.nrt;
    p.nks = p.wsty ? 38 : 32;
    p.rb = p.nrt >= 4 ? 4 : p.nrt;
    p.tbk = 32 / p.rb;
    static bool done[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_ace_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4::LDS_BYTES);
        if (e != hipSuccess) return e;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
        done[dev] = true;
    }
    const int grid = p.ntasks < cus[dev] ? p.ntasks : cus[dev];
    hipLaunchKernelGGL(wino4_ace_kernel<0>, dim3(grid), dim3(512), wino4::LDS_BYTES, s, p);
    return hipGetLastError();
}

// one thread = one 16-byte unit (idx, lane) of a style image: fragments a = 4 idx .. 4 idx + 3 = 36 m + xi of GEMM row 16 m + (lane & 15),
// input "channel" = label j = 4 s + (lane >> 4) (j >= 19, and the sixth image s = 5: zeros)
__global__ __launch_bounds__(256) void wino4_style_pack_kernel(const float* __restrict__ lut, float* __restrict__ wsty, int B, int C, int nrt) {
    const long long n = (long long)B * nrt * 6 * 1152;
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    const int lane = (int)(i & 63), idx = (int)((i >> 6) % 18), s = (int)((i / 1152) % 6);
    const int rt = (int)((i / (1152 * 6)) % nrt), b = (int)(i / (1152LL * 6 * nrt));
    const int j = 4 * s + (lane >> 4);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const int m = (4 * idx) / 36;                    // (36 % 4 == 0: the four fragments of a unit share the half)
    int ch, beta;
    wino4_ace_row(rt * 32 + m * 16 + (lane & 15), ch, beta);
    if (ch < C && j < 19 && s < 5) {
        const float* P = lut + ((long long)(b * 19 + j) * 9) * 2 * C + beta * C + ch;
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = P[(long long)t * 2 * C];
        const float Gm[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int xi = 4 * idx + e - 36 * m, ii = xi / 6, jj = xi % 6;
            float r[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) r[q] = Gm[ii][0] * g[0][q] + Gm[ii][1] * g[1][q] + Gm[ii][2] * g[2][q];
            u[e] = r[0] * Gm[jj][0] + r[1] * Gm[jj][1] + r[2] * Gm[jj][2];
        }
        o = make_float4(u[0], u[1], u[2], u[3]);
    }
    reinterpret_cast<float4*>(wsty)[i] = o;
}
hipError_t wino4_style_pack(const float* lut, float* wsty, int B, int C, hipStream_t s) {
    const int nrt = (C + 15) / 16;
    const long long n = (long long)B * nrt * 6 * 1152;
    hipLaunchKernelGGL(wino4_style_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lut, wsty, B, C, nrt);
    return hipGetLastError();
}


// ---- pre-transformed input route (conv_wino4v.h) ---------------------------------------------------------------------------------------
static hipError_t wino4v_device(int& cus) {
    static bool done[64] = {};
    static int ncu[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4v_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4v::LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4v_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4v::LDS_BYTES);
        if (e != hipSuccess) return e;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        ncu[dev] = v;
        done[dev] = true;
    }
    cus = ncu[dev];
    return hipSuccess;
}
hipError_t wino4v_pack(const Wino4vPackParams& p, hipStream_t s) {
    if (!p.in || !p.v || p.H % 32 || p.W % 32 || p.H < 32 || p.W < 32 || p.nks <= 0 || p.pitch % 4 || p.xoff % 4 || (p.reflect && p.padded)) return hipErrorInvalidValue;
    const long long blocks = (long long)p.B * (p.H / 32) * (p.W / 32) * p.nks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wino4v_pack_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}
hipError_t conv_wino4v_plain(Wino4Params p, hipStream_t s) {
    if (!wino4_supported(p.H, p.W, p.Cin) || !p.v || !p.wpk || !p.out) return hipErrorInvalidValue;
    wino4_fill_launch(p);
    if (p.nks & 1) return hipErrorInvalidValue;
    // task order: a group of 32 consecutive tasks (one XCD, one L2) = 8 row tiles x 4 spatial tiles -- a V tile is twice an A image, so fewer
    // spatial tiles per group than conv_wino4.h's 4 x 8 (-DW4V_RB=2 / 4 / 8: level to +1 %, profiles/r06_wino4v_tuning.txt)
#ifndef W4V_RB
#define W4V_RB 8
#endif
    p.rb = p.nrt >= W4V_RB ? W4V_RB : p.nrt;
    p.tbk = 32 / p.rb;
    int cus = 0;
    hipError_t e = wino4v_device(cus);
    if (e != hipSuccess) return e;
    const int grid = p.ntasks < cus ? p.ntasks : cus;
    hipLaunchKernelGGL(wino4v_kernel<0>, dim3(grid), dim3(512), wino4v::LDS_BYTES, s, p);
    return hipGetLastError();
}
hipError_t conv_wino4v_ace(Wino4AceParams p, hipStream_t s) {
    if (!wino4_ace_supported(p.H, p.W, p.C) || !p.v || !p.wpk || !p.out || !p.x || !p.noise) return hipErrorInvalidValue;
    p.nrt = (p.C + 15) / 16;
    p.ntx = p.W / wino4::TS;
    p.nty = p.H / wino4::TS;
    p.ntiles = p.B * p.ntx * p.nty;
    p.ntasks = p.ntiles * p.nrt;
    p.nks = p.wsty ? 38 : 32;
    p.rb = p.nrt >= 4 ? 4 : p.nrt;
    p.tbk = 32 / p.rb;
    int cus = 0;
    hipError_t e = wino4v_device(cus);
    if (e != hipSuccess) return e;
    const int grid = p.ntasks < cus ? p.ntasks : cus;
    hipLaunchKernelGGL(wino4v_kernel<1>, dim3(grid), dim3(512), wino4v::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace chk
