// f16x3 split-operand plain convs (3x3 and 1x1), SH16 input -> f32 C4 output (see conv_sh16.h)
#include "conv_sh16.h"
namespace chk {
template <int KS>
static hipError_t go(const ConvParams& p, hipStream_t s) {
    // dbg bit 64: wave-specialised persistent kernel (measured slower than the 2-blocks-per-CU kernel for the plain
    // epilogue, whose residual loads it cannot hide; kept selectable for profiling)
    if ((p.dbg & 64) && p.W >= 32 && !(p.partial && p.mtiles_hint_small))
        return launch_sh16_ws<KS, 32, 16, 1, EPI_PLAIN>(p, p.Mrows, s);
    if (p.W >= 32) return launch_sh16<KS, 32, 16, 1, EPI_PLAIN>(p, p.Mrows, s);
    if (p.W > 8) return launch_sh16<KS, 16, 16, 2, EPI_PLAIN>(p, p.Mrows, s);
    return launch_sh16<KS, 8, 8, 8, EPI_PLAIN>(p, p.Mrows, s);
}
hipError_t conv_sh16_plain(const ConvParams& p, int KS, hipStream_t s) {
    return KS == 3 ? go<3>(p, s) : (KS == 1 ? go<1>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
