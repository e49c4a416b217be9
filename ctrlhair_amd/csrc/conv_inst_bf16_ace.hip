// single-term bf16 SPADE conv with fused ACE epilogue: the TERMS = 2 instantiations of conv_sh16.h (BASELINE.json configs[4])
#include "conv_sh16.h"
#ifdef CH_ABLATE
#include "conv_sh16_ws2.h"      // experimental kernel, A/B builds only
#endif
namespace chk {
hipError_t conv_bf16_ace(const ConvParams& p, hipStream_t s) { return dispatch_sh16_ace<2>(p, s); }
}  // namespace chk
