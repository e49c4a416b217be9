// SPADE gamma/beta conv + fused ACE epilogue over the compacted boundary pixels (see conv_ace_sparse.h, ace_sparse.h)
#include "conv_ace_sparse.h"
namespace chk {
template <int TH>
static hipError_t launch_ace_sparse(ConvParams p, hipStream_t s) {
    using Cfg = SpCfg<TH>;
    auto kern = conv_ace_sparse_kernel<TH>;
    static bool attr_set[64] = {};                           // per device (a process may own handles on several GPUs)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    p.nchunks = (p.Cin + Cfg::CK - 1) / Cfg::CK;
    p.mtiles = (p.C + 31) / 32;                    // 64-row wave tiles of 32 channels (gamma | beta)
    p.tiles_x = (p.W + 31) / 32;
    p.tiles_y = (p.H + TH - 1) / TH;
    // upper bound of the block tasks (every pixel a boundary pixel): blocks beyond *sp_total return at once
    const long long ntiles = (long long)p.B * p.tiles_x * p.tiles_y;
    int ng, per;
    sparse_groups(TH, p.mtiles, ng, per);
    const long long grid = ntiles * ((ng * p.mtiles + 3) / 4);
    if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), Cfg::LDS_BYTES, s, p);
    return hipGetLastError();
}
hipError_t conv_ace_sparse(const ConvParams& p, int TH, hipStream_t s) {
    if (!p.sp_list || !p.sp_cnt || !p.sp_work || !p.sp_total || p.W < 32) return hipErrorInvalidValue;
    if (TH == 8) return launch_ace_sparse<8>(p, s);
    if (TH == 16) return launch_ace_sparse<16>(p, s);
    return hipErrorInvalidValue;
}
}  // namespace chk
