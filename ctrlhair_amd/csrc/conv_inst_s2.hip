// stride-2 plain-epilogue instantiations of the MFMA conv (Zencoder / shape-encoder / BiSeNet down-sampling)
#include "conv_mfma.h"
namespace chk {
template <int KS>
static hipError_t s2(const ConvParams& p, hipStream_t s) {
    const int rows = p.Mrows;
    if (p.W >= 32) {
        if (rows <= 64) return launch_conv<KS, 2, 1, 32, 16, 1, CK_S2, EPI_PLAIN>(p, rows, s);
        return launch_conv<KS, 2, 2, 32, 8, 1, CK_S2, EPI_PLAIN>(p, rows, s);
    }
    if (p.W > 8) return launch_conv<KS, 2, 2, 16, 16, 1, CK_S2, EPI_PLAIN>(p, rows, s);
    // <= 4x4 outputs (deep shape-encoder layers: weight streaming): 4 M-waves x (4x4 px x 8 samples) instead of 8x8 px tiles
    if (p.W <= 4 && p.H <= 4 && rows >= 256) return launch_conv<KS, 2, 4, 4, 4, 8, CK_S2, EPI_PLAIN>(p, rows, s);
    return launch_conv<KS, 2, 2, 8, 8, 4, CK_S2, EPI_PLAIN>(p, rows, s);
}
hipError_t conv_plain_s2(const ConvParams& p, int KS, hipStream_t s) {
    switch (KS) {
        case 1: return s2<1>(p, s);
        case 3: return s2<3>(p, s);
        case 4: return s2<4>(p, s);
        default: return hipErrorInvalidValue;
    }
}
}  // namespace chk
