// f16x3 stride-2 convs (space-to-depth view, see conv_sh16.h S2D): shape encoder, Zencoder down-sampling, BiSeNet
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_s2d(const ConvParams& p, int KS, hipStream_t s) {
    return KS == 2 ? dispatch_sh16_s2d<2, false>(p, s) : hipErrorInvalidValue;
}
hipError_t conv_sh16_s2d_c4(const ConvParams& p, int KS, hipStream_t s) {
    if (p.Mrows % 4 != 0) return hipErrorInvalidValue;
    return KS == 2 ? dispatch_sh16_s2d<2, true>(p, s) : (KS == 1 ? dispatch_sh16_s2d<1, true>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
