// f16x3 convs over f32 C4 inputs (split while staged): instantiations used by the BiSeNet trunk (see conv_sh16.h, INC4)
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_plain_c4(const ConvParams& p, int KS, hipStream_t s) {
    if (p.Cin % 16 != 0 || p.Mrows % 4 != 0) return hipErrorInvalidValue;
    return KS == 3 ? dispatch_sh16_plain_c4<3>(p, s) : (KS == 1 ? dispatch_sh16_plain_c4<1>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
