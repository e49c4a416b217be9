// conv_inst_wino4.hip -- instantiation + launcher of the Winograd F(4x4,3x3) exact-f32 MFMA kernel (conv_wino4.h)
#include "conv_wino4.h"

namespace chk {

hipError_t conv_wino4_plain(Wino4Params p, hipStream_t s) {
    if (!wino4_supported(p.H, p.W, p.Cin) || !p.in || !p.wpk || !p.out) return hipErrorInvalidValue;
    wino4_fill_launch(p);
    static bool done[64] = {};
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_plain_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, wino4::LDS_BYTES);
        if (e != hipSuccess) return e;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
        done[dev] = true;
    }
    const int grid = p.ntasks < cus[dev] ? p.ntasks : cus[dev];
    hipLaunchKernelGGL(wino4_plain_kernel<0>, dim3(grid), dim3(512), wino4::LDS_BYTES, s, p);
    return hipGetLastError();
}

}  // namespace chk
