// f16x3 split-operand SPADE conv with fused ACE epilogue (see conv_sh16.h)
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s) {
    if (p.act > ACT_RELU) return hipErrorInvalidValue;   // the ACE epilogue implements none / leaky / relu only
    const int rows = ((p.C + 31) / 32) * 64;
    // dbg bit 64 forces the wave-specialised persistent kernel, bit 128 forbids it; default: layers with at least two
    // rounds of tiles per CU (its loaders then hide every tile's prologue behind the previous tile's epilogue)
    const long long ntiles = (long long)(rows / 64) * ((p.W + 31) / 32) * ((p.H + 15) / 16) * p.B;
    const bool ws_ok = p.W >= 32 && p.Cin >= 48;
    if (ws_ok && ((p.dbg & 64) || (!(p.dbg & 128) && ntiles >= 512)))
        return launch_sh16_ws<3, 32, 16, 1, EPI_ACE>(p, rows, s);
    if (p.W >= 32) return launch_sh16<3, 32, 16, 1, EPI_ACE>(p, rows, s);
    if (p.W > 8) return launch_sh16<3, 16, 16, 2, EPI_ACE>(p, rows, s);
    return launch_sh16<3, 8, 8, 8, EPI_ACE>(p, rows, s);
}
}  // namespace chk
