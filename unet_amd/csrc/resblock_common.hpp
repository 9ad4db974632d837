// Shared by the residual-block layer kernels (resblock.hip: register-staged, any storage type, optional input
// addend; resblock_dma.hip: the LDS-DMA pipeline for a single input).
#pragma once
#include <hip/hip_runtime.h>

namespace unetmi {

constexpr int kRbC = 64;        // channels
constexpr int kRbTR = 4;        // tile rows

struct ResblockArgs {
    const void* x;
    const void* x2;         // optional addend of the input (x + x2)
    const float* dw_w;      // [64][3][3]
    const float* dw_b;      // [64] or null
    const float* pw_w;      // [64 out][64 in]
    const float* pw_b;      // [64] or null
    const float* alpha;     // PReLU parameter [1]
    const float* scale;     // folded BatchNorm [64]
    const float* shift;
    void* out;
    int N, H, W, residual, act_first;
    int tiles_h, tiles_w, ntiles;
};

// The single-input layer on the LDS-DMA pipeline (resblock_dma.hip), fp32 or bfloat16 storage.  Fills tiles_* /
// ntiles itself.
hipError_t launch_resblock_dma(int dilation_w, hipStream_t s, ResblockArgs a, bool bf16);

}  // namespace unetmi
