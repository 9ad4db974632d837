// Kernels for transposed convolutions with time stride 2: two column phases whose
// tap rows (kw = 3 -> 1 and 2 taps, kw = 5 -> 3 and 2 taps) are unrolled.
#include "conv_dma_kernel.hpp"

namespace unetmi {

bool launch_conv_ng2(int wn, int ntw0, int ntw1, int tpg, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p) {
#define UNETMI_NG2(WN_, A_, B_)                                                             \
    if (wn == WN_ && ntw0 == A_ && ntw1 == B_ && tpg == 4) {                                \
        launch_dma_variant<2, 4, 4 / WN_, WN_, A_, B_>(grid, lds, s, p);             \
        return true;                                                                        \
    }
    UNETMI_NG2(2, 1, 2) UNETMI_NG2(2, 3, 2) UNETMI_NG2(1, 1, 2) UNETMI_NG2(1, 3, 2)
#undef UNETMI_NG2
    // one or two row tiles per wave (small and mid-sized grids; conv_one_row_tile_built names exactly these)
#define UNETMI_NG2_T1(A_, B_)                                                               \
    if (wn == 2 && ntw0 == A_ && ntw1 == B_ && tpg == 1) {                                  \
        launch_dma_variant<2, 1, 2, 2, A_, B_>(grid, lds, s, p);                            \
        return true;                                                                        \
    }                                                                                       \
    if (wn == 2 && ntw0 == A_ && ntw1 == B_ && tpg == 2) {                                  \
        launch_dma_variant<2, 2, 2, 2, A_, B_>(grid, lds, s, p);                            \
        return true;                                                                        \
    }
    UNETMI_NG2_T1(1, 2) UNETMI_NG2_T1(3, 2)
#undef UNETMI_NG2_T1
    // five row tiles per wave (ten output rows of each column phase per workgroup, 160 accumulators)
#define UNETMI_NG2_T5(A_, B_)                                                               \
    if (wn == 2 && ntw0 == A_ && ntw1 == B_ && tpg == 5) {                                  \
        launch_dma_variant<2, 5, 2, 2, A_, B_>(grid, lds, s, p);                            \
        return true;                                                                        \
    }
    UNETMI_NG2_T5(1, 2) UNETMI_NG2_T5(3, 2)
#undef UNETMI_NG2_T5
    return false;
}

}  // namespace unetmi
