// single-term bf16 plain convs (3x3 and 1x1): the TERMS = 2 instantiations of conv_sh16.h (BASELINE.json configs[4])
#include "conv_sh16.h"
namespace chk {
hipError_t conv_bf16_plain(const ConvParams& p, int KS, hipStream_t s) {
    return KS == 3 ? dispatch_sh16_plain<3, 2>(p, s) : (KS == 1 ? dispatch_sh16_plain<1, 2>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
