// Kernels with the tap row unrolled (1, 3, 5 or 7 taps along time), one column phase:
// every plain convolution and the transposed ones with time stride 1.
#include "conv_dma_kernel.hpp"

namespace unetmi {

bool launch_conv_ng1(int wn, int ntw0, int tpg, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p) {
    // five / six row tiles per wave: ten / twelve output rows per workgroup
#define UNETMI_NG1_T6(NTW_)                                                                 \
    if (wn == 2 && ntw0 == NTW_ && tpg == 6) {                                              \
        launch_dma_variant<1, 6, 2, 2, NTW_, 0>(grid, lds, s, p);                           \
        return true;                                                                        \
    }                                                                                       \
    if (wn == 2 && ntw0 == NTW_ && tpg == 5) {                                              \
        launch_dma_variant<1, 5, 2, 2, NTW_, 0>(grid, lds, s, p);                           \
        return true;                                                                        \
    }
    UNETMI_NG1_T6(1) UNETMI_NG1_T6(3) UNETMI_NG1_T6(5) UNETMI_NG1_T6(7)
#undef UNETMI_NG1_T6
#define UNETMI_NG1(WN_, NTW_)                                                               \
    if (wn == WN_ && ntw0 == NTW_ && tpg == 4) {                                            \
        launch_dma_variant<1, 4, 4 / WN_, WN_, NTW_, 0>(grid, lds, s, p);            \
        return true;                                                                        \
    }
    UNETMI_NG1(2, 1) UNETMI_NG1(2, 3) UNETMI_NG1(2, 5) UNETMI_NG1(2, 7)
    UNETMI_NG1(1, 1) UNETMI_NG1(1, 3) UNETMI_NG1(1, 5) UNETMI_NG1(1, 7)
#undef UNETMI_NG1
    // one or two row tiles per wave (small and mid-sized grids; conv_one_row_tile_built names exactly these)
#define UNETMI_NG1_T1(NTW_)                                                                 \
    if (wn == 2 && ntw0 == NTW_ && tpg == 1) {                                              \
        launch_dma_variant<1, 1, 2, 2, NTW_, 0>(grid, lds, s, p);                           \
        return true;                                                                        \
    }                                                                                       \
    if (wn == 2 && ntw0 == NTW_ && tpg == 2) {                                              \
        launch_dma_variant<1, 2, 2, 2, NTW_, 0>(grid, lds, s, p);                           \
        return true;                                                                        \
    }
    UNETMI_NG1_T1(1) UNETMI_NG1_T1(3) UNETMI_NG1_T1(5) UNETMI_NG1_T1(7)
#undef UNETMI_NG1_T1
    return false;
}

}  // namespace unetmi
