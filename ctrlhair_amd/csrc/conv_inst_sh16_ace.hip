// f16x3 split-operand SPADE conv with fused ACE epilogue (see conv_sh16.h)
#include "conv_sh16.h"
#ifdef CH_ABLATE
#include "conv_sh16_ws2.h"      // experimental kernel, A/B builds only
#endif
namespace chk {
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s) {
    return (p.terms == 1 || p.terms == 2) ? conv_h16_ace(p, s) : dispatch_sh16_ace<3>(p, s);
}
}  // namespace chk
