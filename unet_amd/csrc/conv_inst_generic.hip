// Run-time-geometry kernels: the register-staged kernel with two loads per input
// element (skip add; concat whose chunks straddle the two sources) and without, and
// the double-buffered DMA kernel with its tap walk at run time (any kernel size / dilation).
#include "conv_dma_kernel.hpp"

namespace unetmi {

bool launch_conv_staged(int ng, int wn, bool two, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p) {
#define UNETMI_GEN(NG_, WN_, TWO_, BF_)                                                     \
    if (ng == NG_ && wn == WN_ && two == TWO_ && (p.bf16 != 0) == BF_) {                    \
        launch_conv_variant<NG_, 4, 4 / WN_, WN_, TWO_, 0, 0, BF_>(grid, lds, s, p);   \
        return true;                                                                        \
    }
    UNETMI_GEN(1, 1, true, false)  UNETMI_GEN(1, 2, true, false)   // one column phase only: 128 accumulator
                                                       // registers leave no room for register staging
    // bfloat16 storage: the loads widen in registers, so every such layer is register-staged
    UNETMI_GEN(1, 1, false, true) UNETMI_GEN(1, 2, false, true) UNETMI_GEN(1, 1, true, true) UNETMI_GEN(1, 2, true, true)
#undef UNETMI_GEN
    return false;
}

bool launch_conv_dma_generic(int ng, int wn, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p) {
#define UNETMI_GEN(NG_, WN_)                                                                \
    if (ng == NG_ && wn == WN_) {                                                           \
        launch_dma_variant<NG_, 4, 4 / WN_, WN_, 0, 0>(grid, lds, s, p);               \
        return true;                                                                        \
    }
    UNETMI_GEN(1, 1) UNETMI_GEN(1, 2) UNETMI_GEN(2, 1) UNETMI_GEN(2, 2)
#undef UNETMI_GEN
    return false;
}

}  // namespace unetmi
