// 1x1 plain-epilogue and NHWC (swapped-operand) instantiations of the MFMA conv (see conv_mfma.h)
#include "conv_mfma.h"
namespace chk {
hipError_t conv_plain1(const ConvParams& p, hipStream_t s) {
    const int rows = p.Mrows;
    if (p.W >= 32) {
        if (rows <= 64) return launch_conv<1, 1, 1, 32, 16, 1, CK_KS1, EPI_PLAIN>(p, rows, s);
        return launch_conv<1, 1, 2, 32, 8, 1, CK_KS1, EPI_PLAIN>(p, rows, s);
    }
    if (p.W > 8) return launch_conv<1, 1, 2, 16, 16, 1, CK_KS1, EPI_PLAIN>(p, rows, s);
    return launch_conv<1, 1, 2, 8, 8, 4, CK_KS1, EPI_PLAIN>(p, rows, s);
}
hipError_t conv_nhwc1x1(const ConvParams& p, hipStream_t s) {
    return launch_conv<1, 1, 2, 32, 8, 1, CK_KS1, EPI_NHWC>(p, p.Mrows, s);
}
}  // namespace chk
