// f16x3 split-operand plain convs (3x3 and 1x1), SH16 input -> f32 C4 output (see conv_sh16.h)
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_plain(const ConvParams& p, int KS, hipStream_t s) {
    if (p.terms == 1 || p.terms == 2) return conv_h16_plain(p, KS, s);
    return KS == 3 ? dispatch_sh16_plain<3, 3>(p, s) : (KS == 1 ? dispatch_sh16_plain<1, 3>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
