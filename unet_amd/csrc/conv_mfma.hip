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
// all column phases, and NB (32|64) output channels.  Per channel chunk it
// stages the input patch [ci][rows][cols] (zero filled outside the image) and
// that chunk's weights into LDS, then every wave runs taps x channel-pairs
// MFMAs on its 4 accumulator tiles.  Two workgroups per CU (<= 64 KB LDS each)
// let one stage while the other computes.
#include <hip/hip_runtime.h>

#include "conv_plan.hpp"

namespace unetmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* in0; const float* in1; const float* wpacked;
    const float* bias; const float* scale; const float* shift; const float* act_param;
    const float* gate; const float* residual; float* out;
    int N, C0, C1, Ctot, H, W, K, OH, OW;
    int in_mode, act, act_first, act_pc;
    int h_nphase, h_in_step, h_tap_step;
    int h_ntap[kMaxPhase], h_base[kMaxPhase], h_nout[kMaxPhase];
    int w_nphase, w_in_step, w_tap_step;
    int w_ntap[kMaxPhase], w_base[kMaxPhase], w_nout[kMaxPhase];
    int w_colmin;
    int ci_chunk, nchunk, NB, nco_blk, TI;
    int nrows, ncols, pitch, plane, in_tile_floats;
    int wtaps[kMaxPhase];
    unsigned long long w_phase_off[kMaxPhase];
    int nit[kMaxPhase], njt;
    unsigned blk_end[kMaxPhase];      // exclusive prefix end of each row phase's blocks
    unsigned magic_ncols, magic_nrows;  // ceil(2^32 / d) for division of small numbers
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case UNETMI_ACT_LEAKY:    return v >= 0.f ? v : v * slope;
        case UNETMI_ACT_RELU:     return fmaxf(v, 0.f);
        case UNETMI_ACT_SILU:     return v / (1.f + expf(-v));
        case UNETMI_ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
        case UNETMI_ACT_SIGMOID:  return 1.f / (1.f + expf(-v));
        case UNETMI_ACT_TANH:     return tanhf(v);
        case UNETMI_ACT_GELU:     return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default:                  return v;
    }
}

template <int NG, int TPG, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + p.in_tile_floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, j = lane & 31;

    // ---- which tile is this workgroup's -----------------------------------
    unsigned bid = blockIdx.x;
    int a = 0;
    unsigned start = 0;
#pragma unroll
    for (int q = 0; q < kMaxPhase - 1; ++q)
        if (q + 1 < p.h_nphase && bid >= p.blk_end[q]) { a = q + 1; start = p.blk_end[q]; }
    bid -= start;
    const int cb = bid % p.nco_blk; bid /= p.nco_blk;
    const int jt = bid % p.njt;     bid /= p.njt;
    const int it = bid % p.nit[a];  bid /= p.nit[a];
    const int n = bid;
    const int i0 = it * p.TI, j0 = jt * kTileJ;
    const int nth = p.h_ntap[a];
    const int row0 = i0 * p.h_in_step + p.h_base[a];
    const int col0 = j0 * p.w_in_step + p.w_colmin;
    const int cpc = p.ci_chunk >> 1;
    const int chunk_w_floats = cpc * p.wtaps[a] * 2 * p.NB;

    f32x16 acc[NG][TPG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.f;

    // per-lane LDS offsets of the B operand (input patch) for each accumulator tile
    int ib[NG][TPG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < TPG; ++t)
            ib[g][t] = half * p.plane + ((wm * TPG + t) * p.h_in_step) * p.pitch +
                       j * p.w_in_step + (p.w_base[g] - p.w_colmin);
    const int wlane = half * p.NB + wn * 32 + j;

    const int in_elems = p.ci_chunk * p.nrows * p.ncols;
    const size_t HW = (size_t)p.H * p.W;
    const float* wsrc_base = p.wpacked + p.w_phase_off[a] + (size_t)cb * p.nchunk * chunk_w_floats;

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        __syncthreads();   // every wave is done reading the previous chunk
        // ---- stage the input patch: [ci][row][col], zero outside -------------
        const int ci_base = chunk * p.ci_chunk;
#pragma unroll 4
        for (int e = tid; e < in_elems; e += 256) {
            const int rr = __umulhi((unsigned)e, p.magic_ncols);
            const int c = e - rr * p.ncols;
            const int cl = (p.nrows == 1) ? rr : (int)__umulhi((unsigned)rr, p.magic_nrows);
            const int r = rr - cl * p.nrows;
            const int ci = ci_base + cl;
            const int grow = row0 + r, gcol = col0 + c;
            float v = 0.f;
            if (ci < p.Ctot && (unsigned)grow < (unsigned)p.H && (unsigned)gcol < (unsigned)p.W) {
                const size_t sp = (size_t)grow * p.W + gcol;
                if (p.in_mode == UNETMI_IN_CONCAT && ci >= p.C0)
                    v = p.in1[((size_t)n * p.C1 + (ci - p.C0)) * HW + sp];
                else {
                    v = p.in0[((size_t)n * p.C0 + ci) * HW + sp];
                    if (p.in_mode == UNETMI_IN_ADD) v += p.in1[((size_t)n * p.C0 + ci) * HW + sp];
                }
            }
            s_in[cl * p.plane + r * p.pitch + c] = v;
        }
        // ---- stage this chunk's weights (already in read order) ----------------
        {
            const float4* src = reinterpret_cast<const float4*>(wsrc_base + (size_t)chunk * chunk_w_floats);
            float4* dst = reinterpret_cast<float4*>(s_w);
            const int n4 = chunk_w_floats >> 2;
#pragma unroll 4
            for (int e = tid; e < n4; e += 256) dst[e] = src[e];
        }
        __syncthreads();

        // ---- MFMA over (channel pair, column phase, tap row, tap col) ----------
        const float* wp = s_w + wlane;
        for (int cp = 0; cp < cpc; ++cp) {
            const int cpoff = cp * 2 * p.plane;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int ntw = p.w_ntap[g];
                for (int th = 0; th < nth; ++th) {
                    const int roff = cpoff + th * p.h_tap_step * p.pitch;
                    for (int tw = 0; tw < ntw; ++tw) {
                        const float wa = *wp;
                        wp += 2 * p.NB;
                        const int off = roff + tw * p.w_tap_step;
                        float xb[TPG];
#pragma unroll
                        for (int t = 0; t < TPG; ++t) xb[t] = s_in[ib[g][t] + off];
#pragma unroll
                        for (int t = 0; t < TPG; ++t)
                            acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb[t], acc[g][t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue ---------------------------------------------------------------
    const int co_base = cb * p.NB + wn * 32 + 4 * half;
    const size_t OHW = (size_t)p.OH * p.OW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co_base + (r & 3) + 8 * (r >> 2);
        if (co >= p.K) continue;
        const float b = p.bias[co], sc = p.scale[co], sf = p.shift[co];
        const float slope = p.act_param ? p.act_param[p.act_pc ? co : 0] : 0.f;
        const float gt = p.gate ? p.gate[(size_t)n * p.K + co] : 1.f;
        const size_t chan = ((size_t)n * p.K + co) * OHW;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int jj = j0 + j;
            const int ow = jj * p.w_nphase + g;
            const bool col_ok = jj < p.w_nout[g];
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                const int i = i0 + wm * TPG + t;
                if (i >= p.h_nout[a] || !col_ok) continue;
                const int oh = i * p.h_nphase + a;
                float v = acc[g][t][r] + b;
                if (p.act_first) v = apply_act(v, p.act, slope) * sc + sf;
                else             v = apply_act(v * sc + sf, p.act, slope);
                v *= gt;
                const size_t o = chan + (size_t)oh * p.OW + ow;
                if (p.residual) v += p.residual[o];
                p.out[o] = v;
            }
        }
    }
}

static unsigned magic_u32(int d) {
    return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d);
}

// Launches the layer.  Returns hipSuccess or the launch error.
hipError_t launch_conv_mfma(const UnetmiConvDesc& d, const ConvPlan& pl,
                            const float* in0, const float* in1, const float* wpacked,
                            const float* bias, const float* scale, const float* shift,
                            const float* act_param, const float* gate, const float* residual,
                            float* out, hipStream_t stream) {
    ConvArgs p;
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
    p.ci_chunk = pl.ci_chunk; p.nchunk = pl.nchunk; p.NB = pl.NB; p.nco_blk = pl.nco_blk; p.TI = pl.TI;
    p.nrows = pl.nrows; p.ncols = pl.ncols; p.pitch = pl.pitch; p.plane = pl.plane;
    p.in_tile_floats = pl.in_tile_floats; p.njt = pl.njt;
    p.magic_ncols = magic_u32(pl.ncols); p.magic_nrows = magic_u32(pl.nrows);
    if (total == 0) return hipSuccess;

    const dim3 grid(total), block(256);
    const size_t lds = pl.lds_bytes;
    if (pl.WN == 2) {
        if (pl.NG == 1) hipLaunchKernelGGL((conv_mfma_kernel<1, 4, 2, 2>), grid, block, lds, stream, p);
        else            hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 2, 2>), grid, block, lds, stream, p);
    } else {
        if (pl.NG == 1) hipLaunchKernelGGL((conv_mfma_kernel<1, 4, 4, 1>), grid, block, lds, stream, p);
        else            hipLaunchKernelGGL((conv_mfma_kernel<2, 2, 4, 1>), grid, block, lds, stream, p);
    }
    return hipGetLastError();
}

}  // namespace unetmi
