// single-term f16 plain convs (3x3 and 1x1): the TERMS = 1 instantiations of conv_sh16.h
#include "conv_sh16.h"
namespace chk {
hipError_t conv_h16_plain(const ConvParams& p, int KS, hipStream_t s) {
    if (p.terms == 2) return conv_bf16_plain(p, KS, s);
    return KS == 3 ? dispatch_sh16_plain<3, 1>(p, s) : (KS == 1 ? dispatch_sh16_plain<1, 1>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
