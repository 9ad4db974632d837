// Implicit-GEMM (transposed) convolution for gfx950 on the exact-fp32 matrix
// instruction v_mfma_f32_32x32x2_f32, with the layer's whole epilogue fused:
// bias, eval-BatchNorm affine, activation (either order), per-(n,k) gate and
// residual add; and with the decoder's channel concat (or ResUNet's skip add)
// folded into the input read, so the concatenated tensor never exists.
//
// GEMM view (per output phase):  D[k][pixel] = sum_{ci,tap} Wt[k][ci,tap] * X[ci,tap][pixel]
//   A operand = weights  (row = output channel, on the lane: l & 31)
//   B operand = input    (col = output pixel along time, on the lane: l & 31)
//   lane half (l >> 5) selects which of two input channels the instruction's
//   K = 2 step covers, so channels are consumed in pairs.
//   D: lane holds ONE pixel and 16 output channels -> every store instruction
//   writes 2 x 128 B contiguous runs along time (NCHW, time innermost).
//
// A workgroup (4 waves) owns: one row phase, TI phase-rows x 32 phase-columns x
// all column phases, and NB (32|64) output channels.  Channels are consumed in
// chunks (a concatenated input is chunked per source tensor, so a chunk streams
// from one tensor).  Two kernels share the MFMA loop and the epilogue:
//   * conv_dma_kernel (conv_dma_kernel.hpp), the default: the LDS image of a chunk
//     (input patch [ci][rows][cols], zero outside the image, plus that chunk's
//     weights, already in read order) is double-buffered and filled by LDS-DMA
//     (buffer_load ... lds), one DMA instruction issued per tap from inside the
//     MFMA loop, one barrier per chunk.  The image size is chosen per layer so
//     that three (25 KB images) or two (38 KB) workgroups share a compute unit.
//   * conv_mfma_kernel (conv_kernel.hpp), register-staged: the next chunk's global
//     loads are in flight in registers while the current chunk computes and are
//     written to LDS between two barriers.  It serves the summed two-source input
//     (the loads of both tensors are added on the way into LDS) and bfloat16
//     storage (the loads are widened in registers).
#include "conv_kernel.hpp"

namespace unetmi {

// Developer probes (tools/conv_probe.hip builds with -DUNETMI_ABLATE); the library itself has no mutable state.
#ifdef UNETMI_ABLATE
int g_ablate = 0;
unsigned long long* g_stamps = nullptr;
int g_force_staged = 0;  // 1: always use the register-staged kernel
int g_xcd_remap = 0;     // 1: one XCD walks a contiguous run of tiles
#else
constexpr int g_force_staged = 0;
constexpr int g_xcd_remap = 0;
#endif

static unsigned magic_u32(int d) {
    return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d);
}

// Partials per channel that a launch with a statistics pointer writes: one per tile of an output-channel block.
int conv_stats_slots(const UnetmiConvDesc& d, const ConvPlan& pl) {
    long long total = 0;
    for (int q = 0; q < pl.h.nphase && q < kMaxPhase; ++q) total += (long long)pl.nit[q] * pl.njt * d.N;
    return (int)total;
}

// Launches the layer.  Returns hipSuccess or the launch error.
hipError_t launch_conv_mfma(const UnetmiConvDesc& d, const ConvPlan& pl,
                            const float* in0, const float* in1, const float* wpacked,
                            const float* bias, const float* scale, const float* shift,
                            const float* act_param, const float* gate, const float* residual,
                            float* out, hipStream_t stream, double* stats, int stats_S) {
    ConvArgs p;
    p.stats = stats; p.stats_S = stats_S;
    p.in0 = in0; p.in1 = in1; p.wpacked = wpacked; p.bias = bias; p.scale = scale; p.shift = shift;
    p.act_param = act_param; p.gate = gate; p.residual = residual; p.out = out;
    p.N = d.N; p.C0 = d.C0; p.C1 = d.C1; p.Ctot = pl.Ctot; p.H = d.H; p.W = d.W; p.K = d.K;
    p.OH = d.OH; p.OW = d.OW;
    p.in_mode = d.in_mode; p.act = d.act; p.act_first = d.act_first; p.act_pc = d.act_param_per_channel;
    p.h_nphase = pl.h.nphase; p.h_in_step = pl.h.in_step; p.h_tap_step = pl.h.tap_step;
    p.w_nphase = pl.w.nphase; p.w_in_step = pl.w.in_step; p.w_tap_step = pl.w.tap_step;
    unsigned total = 0;
    for (int q = 0; q < kMaxPhase; ++q) {
        p.h_ntap[q] = pl.h.ntap[q]; p.h_base[q] = pl.h.base[q]; p.h_nout[q] = pl.h.nout[q];
        p.w_ntap[q] = pl.w.ntap[q]; p.w_base[q] = pl.w.base[q]; p.w_nout[q] = pl.w.nout[q];
        p.wtaps[q] = pl.wtaps[q]; p.w_phase_off[q] = pl.w_phase_off[q]; p.nit[q] = std::max(pl.nit[q], 1);
        if (q < pl.h.nphase) total += (unsigned)pl.nit[q] * pl.njt * d.N * pl.nco_blk;
        p.blk_end[q] = total;
    }
    p.w_colmin = pl.w_colmin;
    p.ci_chunk = pl.ci_chunk; p.nchunk = pl.nchunk; p.nchunk0 = pl.nchunk0; p.NB = pl.NB; p.nco_blk = pl.nco_blk; p.TI = pl.TI;
    p.nrows = pl.nrows; p.ncols = pl.ncols; p.plane = pl.plane;
    p.in_tile_floats = pl.in_tile_floats; p.w_tile_floats = pl.w_tile_floats; p.njt = pl.njt;
    p.magic_ncols = magic_u32(pl.ncols); p.magic_nrows = magic_u32(pl.nrows);
    // One XCD walks a contiguous run of tiles when the layer has a single output phase (plain convolutions and
    // stride-1 transposed ones): neighbouring tiles' shared halo then meets in that XCD's L2 (headline step
    // 36.7 -> 36.3 ms).  With two row phases the same order costs time (dec_6: +0.7 ms), so those keep the
    // round-robin order.
    p.xcd_remap = g_xcd_remap || (pl.NG == 1 && pl.h.nphase == 1);
    p.bf16 = d.reserved[0] == UNETMI_DTYPE_BF16;
#ifdef UNETMI_ABLATE
    p.ablate = g_ablate;
    p.stamps = g_stamps;
#endif
    if (total == 0) return hipSuccess;

    // two loads per input element: the skip add
    const bool two = needs_two_loads(d, pl);
    bool done = false;
    if (p.bf16) {
        done = launch_conv_staged(pl.NG, pl.WN, two, total, pl.lds_bytes, stream, p);
    } else if (two || g_force_staged) {
        done = launch_conv_staged(pl.NG, pl.WN, true, total, pl.lds_bytes, stream, p);
    } else {
        const size_t lds = pl.lds_bytes_dma;
        if (pl.w.tap_step == 1) {          // unrolled tap rows with immediate LDS offsets
            if (pl.NG == 1) done = launch_conv_ng1(pl.WN, pl.w.ntap[0], pl.TPG, total, lds, stream, p);
            else            done = launch_conv_ng2(pl.WN, pl.w.ntap[0], pl.w.ntap[1], pl.TPG, total, lds, stream, p);
        }
        if (!done && pl.TPG == 4) done = launch_conv_dma_generic(pl.NG, pl.WN, total, lds, stream, p);
    }
    if (!done) return hipErrorInvalidConfiguration;
    return hipGetLastError();
}

}  // namespace unetmi
