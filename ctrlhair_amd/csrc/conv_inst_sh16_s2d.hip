// f16x3 stride-2 convs (space-to-depth view, see conv_sh16.h S2D): shape encoder, Zencoder down-sampling, BiSeNet;
// and the Zencoder's ConvTranspose2d as a 2x2-tap conv with a depth-to-space store (D2S)
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_s2d(const ConvParams& p, int KS, hipStream_t s) {
    return KS == 2 ? dispatch_sh16_s2d<2, false>(p, s) : hipErrorInvalidValue;
}
hipError_t conv_sh16_d2s(const ConvParams& p, hipStream_t s) {
    if (p.W < 32 || p.Mrows % 16 != 0 || p.partial || p.res || p.in2 || p.act > ACT_RELU) return hipErrorInvalidValue;
    return launch_sh16<2, 32, 16, 1, EPI_PLAIN, 3, false, false, false, true>(p, p.Mrows, s);
}
hipError_t conv_sh16_s2d_c4(const ConvParams& p, int KS, hipStream_t s) {
    if (p.Mrows % 4 != 0) return hipErrorInvalidValue;
    return KS == 2 ? dispatch_sh16_s2d<2, true>(p, s) : (KS == 1 ? dispatch_sh16_s2d<1, true>(p, s) : hipErrorInvalidValue);
}
}  // namespace chk
