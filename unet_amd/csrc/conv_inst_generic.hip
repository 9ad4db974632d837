// Generic kernels: tap geometry walked at run time (any kernel size / dilation),
// and the two-loads-per-element forms (skip add, concat straddling a chunk).
#include "conv_kernel.hpp"

namespace unetmi {

bool launch_conv_generic(int ng, int wn, bool two, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p) {
#define UNETMI_GEN(NG_, WN_, TWO_)                                                          \
    if (ng == NG_ && wn == WN_ && two == TWO_) {                                            \
        launch_conv_variant<NG_, 4 / NG_, 4 / WN_, WN_, TWO_, 0, 0>(grid, lds, s, p);        \
        return true;                                                                        \
    }
    UNETMI_GEN(1, 1, false) UNETMI_GEN(1, 2, false) UNETMI_GEN(2, 1, false) UNETMI_GEN(2, 2, false)
    UNETMI_GEN(1, 1, true)  UNETMI_GEN(1, 2, true)  UNETMI_GEN(2, 1, true)  UNETMI_GEN(2, 2, true)
#undef UNETMI_GEN
    return false;
}

}  // namespace unetmi
