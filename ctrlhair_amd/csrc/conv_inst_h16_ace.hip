// single-term f16 SPADE conv with fused ACE epilogue: the TERMS = 1 instantiations of conv_sh16.h
#include "conv_sh16.h"
#ifdef CH_ABLATE
#include "conv_sh16_ws2.h"      // experimental kernel, A/B builds only
#endif
namespace chk {
hipError_t conv_h16_ace(const ConvParams& p, hipStream_t s) { return p.terms == 2 ? conv_bf16_ace(p, s) : dispatch_sh16_ace<1>(p, s); }
}  // namespace chk
