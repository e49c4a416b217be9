// f16x3 split-operand SPADE conv with fused ACE epilogue (see conv_sh16.h)
#include "conv_sh16.h"
namespace chk {
hipError_t conv_sh16_ace(const ConvParams& p, hipStream_t s) {
    if (p.act > ACT_RELU) return hipErrorInvalidValue;   // the f16x3 epilogues implement none / leaky / relu only
    const int rows = ((p.C + 31) / 32) * 64;
    if (p.gen_table) {   // input (SPADE hidden activations) generated in-kernel from the label map
        if (p.W >= 32) return launch_sh16_gen<32, 16, 1>(p, rows, s);
        if (p.W > 8) return launch_sh16_gen<16, 16, 2>(p, rows, s);
        return launch_sh16_gen<8, 8, 8>(p, rows, s);
    }
    if (p.dbg & 64) {   // v3: wave-specialised persistent kernel (loaders + consumers), one block per CU
        if (p.W >= 32 && p.Cin >= 48) return launch_sh16v3<3, 32, 16, 1, EPI_ACE>(p, rows, s);
    }
    if (p.zeros) {   // v2: LDS-DMA ring, one block per CU
        if (p.W >= 32) return launch_sh16v2<3, 32, 16, 1, EPI_ACE>(p, rows, s);
        if (p.W > 8) return launch_sh16v2<3, 16, 16, 2, EPI_ACE>(p, rows, s);
    }
    if (p.W >= 32) return launch_sh16<3, 32, 16, 1, EPI_ACE>(p, rows, s);
    if (p.W > 8) return launch_sh16<3, 16, 16, 2, EPI_ACE>(p, rows, s);
    return launch_sh16<3, 8, 8, 8, EPI_ACE>(p, rows, s);
}
}  // namespace chk
