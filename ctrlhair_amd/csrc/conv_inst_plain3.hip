// 3x3 plain-epilogue instantiations of the MFMA conv (see conv_mfma.h)
#include "conv_mfma.h"
namespace chk {
hipError_t conv_plain3(const ConvParams& p, hipStream_t s) {
    const int rows = p.Mrows;
    if (p.in2) {          // 3x3 conv with a fused 1x1 second operand (ResBlock shortcut conv_s folded into conv_1)
        if (p.W < 32 || p.in_mode != IN_DIRECT || p.pad_mode != PAD_ZERO) return hipErrorInvalidValue;
        if (rows <= 64) return launch_conv<3, 1, 1, 32, 16, 1, CK_KS3, EPI_PLAIN, true>(p, rows, s);
        return launch_conv<3, 1, 2, 32, 8, 1, CK_KS3, EPI_PLAIN, true>(p, rows, s);
    }
    if (p.W >= 32) {
        if (rows <= 64) return launch_conv<3, 1, 1, 32, 16, 1, CK_KS3, EPI_PLAIN>(p, rows, s);
        return launch_conv<3, 1, 2, 32, 8, 1, CK_KS3, EPI_PLAIN>(p, rows, s);
    }
    if (p.W > 8) return launch_conv<3, 1, 2, 16, 16, 1, CK_KS3, EPI_PLAIN>(p, rows, s);
    if (p.W <= 4 && p.H <= 4 && rows >= 256) return launch_conv<3, 1, 4, 4, 4, 8, CK_KS3, EPI_PLAIN>(p, rows, s);   // see conv_inst_s2.hip
    return launch_conv<3, 1, 2, 8, 8, 4, CK_KS3, EPI_PLAIN>(p, rows, s);
}
}  // namespace chk
