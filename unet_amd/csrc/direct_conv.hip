// Direct (vector-ALU) convolution for the thin ends of the networks, where the
// work is bandwidth- not matrix-bound: very few output channels (the last
// decoder 96 -> 1, the 1x1 heads, ResUNet's 64 -> 2 output layer) or very few
// input channels (the first encoder 2 -> 32).  Kernel height 1, 3 or 7 along
// frequency, width 1, stride 1 -- the only shapes those layers take.
//
// A thread owns one time column and RPT consecutive frequency rows for KT output
// channels: per input channel it loads RPT+KH-1 values once (rows shared by the
// KH taps stay in registers) and its weights come through the scalar cache
// (they are wave-uniform).  Lanes run along time, so every load and store of a
// wave is one contiguous 256-byte run.  Same fused epilogue and the same
// two-source (concat / add) input read as the MFMA kernel.
#include <hip/hip_runtime.h>

#include "conv_plan.hpp"
#include "io_types.hpp"

namespace unetmi {

struct DirectArgs {
    const float* in0; const float* in1; const float* w;
    const float* bias; const float* scale; const float* shift; const float* act_param;
    const float* gate; const float* residual; float* out;
    int N, C0, C1, Ctot, H, W, K, OH;
    int in_mode, act, act_first, act_pc;
    int pad;        // rows: ih = oh - pad + kh  (transposed layers are packed flipped)
    int kpad;       // padded K of the weight image [ci][kh][kpad]
    int nkt;        // K tiles
};

// The activation is a template parameter of the epilogue: a run-time switch inlined at each of the KT*RPT
// results made the <2,16,*> kernels 24,000 instructions long (190 KB of code against a 64 KB instruction
// cache), and the waves spent their time fetching it.
template <int ACT>
__device__ __forceinline__ float direct_act(float v, float slope) {
    if constexpr (ACT == UNETMI_ACT_LEAKY)         return v >= 0.f ? v : v * slope;
    else if constexpr (ACT == UNETMI_ACT_RELU)     return fmaxf(v, 0.f);
    else if constexpr (ACT == UNETMI_ACT_SILU)     return v * fast_sigmoid(v);
    else if constexpr (ACT == UNETMI_ACT_SOFTPLUS) return fast_softplus(v);
    else if constexpr (ACT == UNETMI_ACT_SIGMOID)  return fast_sigmoid(v);
    else if constexpr (ACT == UNETMI_ACT_TANH)     return fast_tanh(v);
    else if constexpr (ACT == UNETMI_ACT_GELU)     return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    else                                           return v;
}

template <int RPT, int KT, bool BF16, int ACT>
__device__ __forceinline__ void direct_epilogue(const DirectArgs& p, const float (&acc)[KT][RPT], int n, int k0,
                                                int oh0, int ow) {
    const size_t OHW = (size_t)p.OH * p.W;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
        const int co = k0 + kk;
        if (co >= p.K) continue;
        const float b = p.bias[co], sc = p.scale[co], sf = p.shift[co];
        const float slope = p.act_param ? p.act_param[p.act_pc ? co : 0] : 0.f;
        const float gt = p.gate ? p.gate[(size_t)n * p.K + co] : 1.f;
        // act_first: act(v) * sc + sf; otherwise act(v * sc + sf).  Written with one call site for the
        // activation (sc0/sf0 before it, sc1/sf1 after it; the unused pair is the identity 1, 0 -- exact)
        const float sc0 = p.act_first ? 1.f : sc, sf0 = p.act_first ? 0.f : sf;
        const float sc1 = p.act_first ? sc : 1.f, sf1 = p.act_first ? sf : 0.f;
        const size_t chan = ((size_t)n * p.K + co) * OHW;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int oh = oh0 + r;
            if (oh >= p.OH) continue;
            float v = acc[kk][r] + b;
            v = direct_act<ACT>(v * sc0 + sf0, slope) * sc1 + sf1;
            v *= gt;
            const size_t o = chan + (size_t)oh * p.W + ow;
            if (p.residual) v += ld_act<BF16>(p.residual, o);
            st_act<BF16>(p.out, o, v);
        }
    }
}

template <int RPT, int KT, int KH, bool BF16>
__global__ __launch_bounds__(256) void direct_conv_kernel(const DirectArgs p) {
    constexpr int EB = BF16 ? 2 : 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Workgroups go to the 8 XCDs round-robin by their linear id, and each XCD has its own L2.  Tile t of the
    // (sample x output tile, row block, column strip) order -- strips fastest -- goes to the XCD whose contiguous
    // eighth of the tiles holds it, so one XCD walks a contiguous range of tiles: the 128-byte
    // lines that neighbouring 64-column strips share (rows of 509 floats are not line-aligned) and the halo rows
    // that neighbouring row blocks share are then fetched, and partial lines merged, in ONE L2.
    const unsigned tiles = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, q = tiles >> 3, r = tiles & 7u;      // the first r XCDs walk q + 1 tiles, the rest q
    const unsigned t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    const unsigned bx = t % gridDim.x, by = (t / gridDim.x) % gridDim.y, bz = t / (gridDim.x * gridDim.y);
    const int ow = bx * 64 + lane;
    const int oh0 = (by * 4 + wave) * RPT;
    const int n = bz / p.nkt;
    const int k0 = (bz % p.nkt) * KT;
    if (oh0 >= p.OH) return;                       // wave-uniform
    const bool col_ok = ow < p.W;
    const int HW = p.H * p.W;
    constexpr int NX = RPT + KH - 1;

    float acc[KT][RPT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk)
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[kk][r] = 0.f;

    // row validity is the same for every channel
    int roff[NX];
#pragma unroll
    for (int q = 0; q < NX; ++q) {
        const int ih = oh0 - p.pad + q;
        roff[q] = (col_ok && (unsigned)ih < (unsigned)p.H) ? ih * p.W + ow : -1;
    }

    for (int ci = 0; ci < p.Ctot; ++ci) {
        const char* src;
        const char* src2 = nullptr;
        if (p.in_mode == UNETMI_IN_CONCAT && ci >= p.C0)
            src = reinterpret_cast<const char*>(p.in1) + ((size_t)n * p.C1 + (ci - p.C0)) * HW * EB;
        else src = reinterpret_cast<const char*>(p.in0) + ((size_t)n * p.C0 + ci) * HW * EB;
        if (p.in_mode == UNETMI_IN_ADD) src2 = reinterpret_cast<const char*>(p.in1) + ((size_t)n * p.C0 + ci) * HW * EB;
        float x[NX];
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int o = roff[q] < 0 ? 0 : roff[q];
            float v = ld_act<BF16>(src, o);
            if (src2) v += ld_act<BF16>(src2, o);
            x[q] = roff[q] < 0 ? 0.f : v;
        }
        const float* wrow = p.w + (size_t)ci * KH * p.kpad + k0;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                const float wv = wrow[kh * p.kpad + kk];
#pragma unroll
                for (int r = 0; r < RPT; ++r) acc[kk][r] = fmaf(x[r + kh], wv, acc[kk][r]);
            }
    }

    if (!col_ok) return;
    switch (p.act) {
#define UNETMI_DIRECT_EPI(A_) case A_: direct_epilogue<RPT, KT, BF16, A_>(p, acc, n, k0, oh0, ow); break;
        UNETMI_DIRECT_EPI(UNETMI_ACT_LEAKY) UNETMI_DIRECT_EPI(UNETMI_ACT_RELU) UNETMI_DIRECT_EPI(UNETMI_ACT_SILU)
        UNETMI_DIRECT_EPI(UNETMI_ACT_SOFTPLUS) UNETMI_DIRECT_EPI(UNETMI_ACT_SIGMOID) UNETMI_DIRECT_EPI(UNETMI_ACT_TANH)
        UNETMI_DIRECT_EPI(UNETMI_ACT_GELU)
#undef UNETMI_DIRECT_EPI
        default: direct_epilogue<RPT, KT, BF16, UNETMI_ACT_NONE>(p, acc, n, k0, oh0, ow); break;
    }
}

hipError_t launch_direct_conv(const UnetmiConvDesc& d, const ConvPlan& pl,
                              const float* in0, const float* in1, const float* wpacked,
                              const float* bias, const float* scale, const float* shift,
                              const float* act_param, const float* gate, const float* residual,
                              float* out, hipStream_t stream) {
    DirectArgs p;
    p.in0 = in0; p.in1 = in1; p.w = wpacked; p.bias = bias; p.scale = scale; p.shift = shift;
    p.act_param = act_param; p.gate = gate; p.residual = residual; p.out = out;
    p.N = d.N; p.C0 = d.C0; p.C1 = d.C1; p.Ctot = pl.Ctot; p.H = d.H; p.W = d.W; p.K = d.K; p.OH = d.OH;
    p.in_mode = d.in_mode; p.act = d.act; p.act_first = d.act_first; p.act_pc = d.act_param_per_channel;
    p.pad = pl.direct_pad; p.kpad = pl.direct_kpad; p.nkt = pl.direct_kpad / pl.direct_kt;
    const int rpt = pl.direct_rpt, kt = pl.direct_kt, kh = d.kh;
    const bool bf16 = d.reserved[0] == UNETMI_DTYPE_BF16;
    const dim3 grid((d.W + 63) / 64, (d.OH + 4 * rpt - 1) / (4 * rpt), (unsigned)d.N * p.nkt);
    if (grid.z > 65535u * 16u) return hipErrorInvalidConfiguration;
#define UNETMI_DIRECT(RPT_, KT_, KH_)                                                        \
    if (rpt == RPT_ && kt == KT_ && kh == KH_) {                                             \
        if (bf16) hipLaunchKernelGGL((direct_conv_kernel<RPT_, KT_, KH_, true>), grid, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((direct_conv_kernel<RPT_, KT_, KH_, false>), grid, dim3(256), 0, stream, p); \
        return hipGetLastError();                                                            \
    }
    UNETMI_DIRECT(8, 1, 1) UNETMI_DIRECT(8, 1, 3) UNETMI_DIRECT(8, 1, 7)
    UNETMI_DIRECT(16, 1, 1) UNETMI_DIRECT(16, 1, 3) UNETMI_DIRECT(16, 1, 7)
    UNETMI_DIRECT(8, 2, 1) UNETMI_DIRECT(8, 2, 3) UNETMI_DIRECT(8, 2, 7)
    UNETMI_DIRECT(2, 16, 1) UNETMI_DIRECT(2, 16, 3) UNETMI_DIRECT(2, 16, 7)
    UNETMI_DIRECT(4, 16, 1) UNETMI_DIRECT(4, 16, 3) UNETMI_DIRECT(4, 16, 7)
    UNETMI_DIRECT(2, 32, 1) UNETMI_DIRECT(2, 32, 3) UNETMI_DIRECT(2, 32, 7)
#undef UNETMI_DIRECT
    return hipErrorInvalidConfiguration;
}

}  // namespace unetmi
