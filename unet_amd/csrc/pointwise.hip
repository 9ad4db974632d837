// Bandwidth-bound kernels around the convolutions: depthwise (grouped) 3x3 dilated
// convolution of the residual blocks, LayerNorm over the frequency axis, and the
// mask / ratio-filter application over the (complex) STFT.  Lanes run along time
// (the contiguous axis), every wave access is one contiguous run, no LDS.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

// out[n,c,h,w] = b[c] + sum_{kh,kw} (in [+ in2])[n,c,h-ph+kh*dh, w-pw+kw*dw] * wt[c,kh,kw]   (stride 1, groups = C)
template <int KH, int KW, bool BF16>
__global__ __launch_bounds__(256) void dwconv_kernel(const void* __restrict__ in, const void* __restrict__ in2,
                                                     const float* __restrict__ wt,
                                                     const float* __restrict__ bias, void* __restrict__ out,
                                                     int C, int H, int W, int ph, int pw, int dh, int dw) {
    const int w = blockIdx.x * 256 + threadIdx.x;
    const int h = blockIdx.y;
    const int nc = blockIdx.z;                    // n*C + c
    if (w >= W) return;
    const int c = nc % C;
    const size_t base = (size_t)nc * H * W;
    float k[KH * KW];
#pragma unroll
    for (int i = 0; i < KH * KW; ++i) k[i] = wt[c * KH * KW + i];   // wave-uniform -> scalar loads
    float acc = bias ? bias[c] : 0.f;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = h - ph + kh * dh;
        if ((unsigned)ih >= (unsigned)H) continue;                  // wave-uniform
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = w - pw + kw * dw;
            float v = 0.f;
            if ((unsigned)iw < (unsigned)W) {
                v = ld_act<BF16>(in, base + (size_t)ih * W + iw);
                if (in2) v += ld_act<BF16>(in2, base + (size_t)ih * W + iw);      // optional addend (x + skip)
            }
            acc = fmaf(v, k[kh * KW + kw], acc);
        }
    }
    st_act<BF16>(out, base + (size_t)h * W + w, acc);
}

// LayerNorm over F of x[N,C,F,T] (biased variance), written to channels
// [c_off, c_off + C) of out[N, C_out, F, T].  One thread per (n, c, t) column: two
// passes over its F values, each a coalesced row read across the wave.
template <bool BF16>
__global__ __launch_bounds__(256) void layernorm_f_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, void* __restrict__ out,
                                                          int C, int F, int T, int C_out, int c_off, float eps) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int nc = blockIdx.y;
    if (t >= T) return;
    const int n = nc / C, c = nc % C;
    const size_t col = (size_t)nc * F * T + t;
    float mean = 0.f;
    for (int f = 0; f < F; ++f) mean += ld_act<BF16>(x, col + (size_t)f * T);
    mean /= (float)F;
    float var = 0.f;
    for (int f = 0; f < F; ++f) { const float d = ld_act<BF16>(x, col + (size_t)f * T) - mean; var = fmaf(d, d, var); }
    const float rstd = rsqrtf(var / (float)F + eps);
    const size_t dst = ((size_t)n * C_out + c_off + c) * F * T + t;
    for (int f = 0; f < F; ++f)
        st_act<BF16>(out, dst + (size_t)f * T, (ld_act<BF16>(x, col + (size_t)f * T) - mean) * rstd * gamma[f] + beta[f]);
}

// The same LayerNorm with ONE read and one write of the tensor, for F <= 4 * RMAX: a workgroup owns a strip of 64
// columns of one (n, c) plane, wave w the rows [w * ceil(F/4), (w+1) * ceil(F/4)), each lane keeps its column's
// values of those rows in registers; the column sums cross the four waves through LDS (mean first, then the
// squared deviations from it: the two-pass variance without a second read).  Rows are wave-uniform, so the only
// per-lane condition is the column, carried as an out-of-range buffer offset.
template <bool BF16, int RMAX>
__global__ __launch_bounds__(256) void layernorm_f_strip_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, void* __restrict__ out,
                                                                int C, int F, int T, int C_out, int c_off, float eps) {
    constexpr int EB = BF16 ? 2 : 4;
    __shared__ float s_red[2][4][64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one XCD walks a contiguous run of (plane, strip) tiles: rows of T floats are not line-aligned, so neighbouring
    // strips share 128-byte lines, which then meet in one L2 instead of two
    const unsigned tiles = gridDim.x * gridDim.y, lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned xcd = lin & 7u, q8 = tiles >> 3, r8 = tiles & 7u;
    const unsigned tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int t = (int)(tile % gridDim.x) * 64 + lane;
    const int nc = (int)(tile / gridDim.x), n = nc / C, c = nc % C;
    const int R = (F + 3) >> 2, f0 = w * R;
    const int plane_bytes = F * T * EB;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(x)) + (size_t)nc * plane_bytes, 0, plane_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<char*>(out) + ((size_t)n * C_out + c_off + c) * plane_bytes, 0, plane_bytes, 0x00020000);
    const int voff = t < T ? t * EB : (int)0x80000000;
    float v[RMAX];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        v[i] = 0.f;
        if (i < R && f0 + i < F) v[i] = buf_ld_act<BF16>(rs_in, voff, (f0 + i) * T * EB);      // wave-uniform condition
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) sum += v[i];
    s_red[0][w][lane] = sum;
    __syncthreads();
    const float mean = (s_red[0][0][lane] + s_red[0][1][lane] + s_red[0][2][lane] + s_red[0][3][lane]) / (float)F;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < RMAX; ++i)
        if (i < R && f0 + i < F) { const float d = v[i] - mean; var = fmaf(d, d, var); }
    s_red[1][w][lane] = var;
    __syncthreads();
    const float rstd = rsqrtf((s_red[1][0][lane] + s_red[1][1][lane] + s_red[1][2][lane] + s_red[1][3][lane]) / (float)F + eps);
#pragma unroll
    for (int i = 0; i < RMAX; ++i)
        if (i < R && f0 + i < F)
            buf_st_act<BF16>((v[i] - mean) * rstd * gamma[f0 + i] + beta[f0 + i], rs_out, voff, (f0 + i) * T * EB);
}

// InstanceNorm2d (per (n, c) plane over H x W, biased variance; optional per-channel affine) followed by the block's
// activation, and optionally a per-(n, c) gate, in one launch: one workgroup per plane, the statistics pass
// (deviations from the plane's first value, so a large mean does not cancel) and the normalising pass both
// stream the plane, which is at most a few hundred KB and stays in L2 between them.
__device__ __forceinline__ float in_act(int act, float u, float p) {
    switch (act) {
        case UNETMI_ACT_LEAKY: return u >= 0.f ? u : p * u;
        case UNETMI_ACT_RELU: return fmaxf(u, 0.f);
        case UNETMI_ACT_SILU: return u * fast_sigmoid(u);
        case UNETMI_ACT_SOFTPLUS: return fast_softplus(u);
        case UNETMI_ACT_SIGMOID: return fast_sigmoid(u);
        case UNETMI_ACT_TANH: return fast_tanh(u);
        case UNETMI_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void instnorm_act_kernel(const void* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ act_param,
                                                           int act_pc, const float* __restrict__ gate,
                                                           void* __restrict__ out, int C, int HW, float eps, int act) {
    __shared__ double s_red[2][4];
    const int nc = blockIdx.x, c = nc % C;
    const size_t base = (size_t)nc * HW;
    const float x0 = ld_act<BF16>(z, base);
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float d = ld_act<BF16>(z, base + i) - x0;
        s0 += d;
        s1 = fmaf(d, d, s1);
    }
    double d0 = s0, d1 = s1;
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_down(d0, o, 64); d1 += __shfl_down(d1, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = d0; s_red[1][threadIdx.x >> 6] = d1; }
    __syncthreads();
    const double t0 = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    const double t1 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    const double md = t0 / HW;
    double var = t1 / HW - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = x0 + (float)md;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = (gamma ? gamma[c] : 1.f) * rstd;
    const float b = (beta ? beta[c] : 0.f) - mean * g;
    const float ap = act_param ? act_param[act_pc ? c : 0] : 0.f;
    const float gt = gate ? gate[nc] : 1.f;
    for (int i = threadIdx.x; i < HW; i += 256)
        st_act<BF16>(out, base + i, in_act(act, fmaf(ld_act<BF16>(z, base + i), g, b), ap) * gt);
}

// nn.LayerNorm(W) on an NCHW tensor normalises every ROW (n, c, h) over its W values, with gamma / beta indexed by the
// position in the row (what an Encoder / Decoder built with norm="LayerNorm" does, UNet_m.py:312-319: nn.LayerNorm(
// out_channels) applied to [B, C, F, T] needs T == out_channels), followed here by the block's activation and gate.
// One wave per row, four rows per workgroup; the row is read twice (statistics, then normalise) and stays in cache.
template <bool BF16>
__global__ __launch_bounds__(256) void rownorm_act_kernel(const void* __restrict__ z, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ act_param,
                                                          int act_pc, const float* __restrict__ gate, void* __restrict__ out,
                                                          int C, int H, int W, long long rows, float eps, int act) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                          // whole waves leave: no barrier in this kernel
    const int lane = threadIdx.x & 63;
    const size_t base = (size_t)row * W;
    const long long nc = row / H;
    const int c = (int)(nc % C);
    const float x0 = ld_act<BF16>(z, base);
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane; i < W; i += 64) {
        const float d = ld_act<BF16>(z, base + i) - x0;
        s0 += d;
        s1 = fmaf(d, d, s1);
    }
    double d0 = s0, d1 = s1;
    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); }
    const double md = d0 / W;
    double var = d1 / W - md * md;
    if (var < 0.0) var = 0.0;
    const float mean = x0 + (float)md;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ap = act_param ? act_param[act_pc ? c : 0] : 0.f;
    const float gt = gate ? gate[nc] : 1.f;
    for (int i = lane; i < W; i += 64) {
        const float xh = (ld_act<BF16>(z, base + i) - mean) * rstd;
        const float u = gamma ? fmaf(xh, gamma[i], beta ? beta[i] : 0.f) : xh;
        st_act<BF16>(out, base + i, in_act(act, u, ap) * gt);
    }
}

// out[n,k,f,t] = (tanh(y[n,k,f,t]) + eps) * x[n,k,f,t], x having Cx >= K channels (first K used)
template <bool BF16>
__global__ __launch_bounds__(256) void tanh_mask_kernel(const void* __restrict__ y, const void* __restrict__ x,
                                                        void* __restrict__ out, int K, int Cx, size_t HW,
                                                        size_t total, float eps) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t sp = i % HW, nk = i / HW;
    const size_t n = nk / K, k = nk % K;
    st_act<BF16>(out, i, (fast_tanh(ld_act<BF16>(y, i)) + eps) * ld_act<BF16>(x, (n * Cx + k) * HW + sp));
}

// Gated-convolution epilogue: z[N,2K,HW] holds the feature conv in channels [0,K) and the mask conv's
// logits in [K,2K) (one double-width convolution launch); out[n,k] = gate[n,k] * act(scale[k] * (sigmoid(m) * c) + shift[k]).
__device__ __forceinline__ float glu_act(int act, float u, float p) {
    switch (act) {
        case UNETMI_ACT_LEAKY: return u >= 0.f ? u : p * u;
        case UNETMI_ACT_RELU: return fmaxf(u, 0.f);
        case UNETMI_ACT_SILU: return u * fast_sigmoid(u);
        case UNETMI_ACT_SOFTPLUS: return fast_softplus(u);
        case UNETMI_ACT_SIGMOID: return fast_sigmoid(u);
        case UNETMI_ACT_TANH: return fast_tanh(u);
        case UNETMI_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) void glu_kernel(const void* __restrict__ z, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, const float* __restrict__ act_param,
                                                  const float* __restrict__ gate, void* __restrict__ out, int K, int HW,
                                                  int act) {
    const int nk = blockIdx.y, n = nk / K, k = nk - n * K;
    const float sc = scale ? scale[k] : 1.f, sh = shift ? shift[k] : 0.f;
    const float ap = act_param ? act_param[0] : 0.f, g = gate ? gate[nk] : 1.f;
    const size_t c_off = ((size_t)n * 2 * K + k) * HW, m_off = c_off + (size_t)K * HW, o_off = (size_t)nk * HW;
    const int base = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 256;
        if (i < HW) {
            const float c = ld_act<BF16>(z, c_off + i), m = ld_act<BF16>(z, m_off + i);
            const float v = c * fast_sigmoid(m);
            st_act<BF16>(out, o_off + i, g * glu_act(act, fmaf(v, sc, sh), ap));
        }
    }
}

// Complex ratio filter: Y[b,n,c,f,t] = sum_{p,q} Hf[b,n,c,p,q,f,t] * X[b,c,f+p-Lf,t+q-Lt]  (zero outside).
// All tensors complex64 viewed as float2; one thread per output element, t innermost.
__global__ __launch_bounds__(256) void crf_apply_kernel(const float2* __restrict__ Hf, const float2* __restrict__ X,
                                                        float2* __restrict__ Y, int N, int C, int PF, int PT,
                                                        int F, int T, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    const int f = (int)((i / T) % F);
    const size_t bnc = i / ((size_t)F * T);          // (b*N + n)*C + c
    const int c = (int)(bnc % C);
    const size_t b = bnc / ((size_t)N * C);
    const int Lf = (PF - 1) / 2, Lt = (PT - 1) / 2;
    const size_t FT = (size_t)F * T;
    const float2* xs = X + (b * C + c) * FT;
    float re = 0.f, im = 0.f;
    for (int p = 0; p < PF; ++p) {
        const int ff = f + p - Lf;
        if ((unsigned)ff >= (unsigned)F) continue;
        for (int q = 0; q < PT; ++q) {
            const int tt = t + q - Lt;
            if ((unsigned)tt >= (unsigned)T) continue;
            const float2 h = Hf[((bnc * PF + p) * PT + q) * FT + (size_t)f * T + t];
            const float2 v = xs[(size_t)ff * T + tt];
            re += h.x * v.x - h.y * v.y;
            im += h.x * v.y + h.y * v.x;
        }
    }
    Y[i] = make_float2(re, im);
}

}  // namespace unetmi

using namespace unetmi;

namespace unetmi { int api_fail(int code, const std::string& msg); }

extern "C" {

int unetmi_dwconv2d_fwd(int N, int C, int H, int W, int kh, int kw, int ph, int pw, int dh, int dw,
                        const void* in, const void* in2, const float* weight, const float* bias, void* out,
                        int dtype, void* stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1) return api_fail(UNETMI_E_BAD_DESC, "dwconv: bad sizes");
    if (2 * ph != dh * (kh - 1) || 2 * pw != dw * (kw - 1))
        return api_fail(UNETMI_E_UNSUPPORTED, "dwconv: only 'same' padding (2p = d(k-1)) is built");
    if (N == 0) return UNETMI_OK;
    if (!in || !weight || !out) return api_fail(UNETMI_E_BAD_ARG, "dwconv: null tensor pointer");
    if ((long long)N * C > 65535ll * 1024) return api_fail(UNETMI_E_UNSUPPORTED, "dwconv: N*C too large");
    const dim3 grid((W + 255) / 256, H, N * C), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool bf = dtype == UNETMI_DTYPE_BF16;
#define UNETMI_DW(KH_, KW_)                                                                                              \
    if (kh == KH_ && kw == KW_) {                                                                                        \
        if (bf) hipLaunchKernelGGL((dwconv_kernel<KH_, KW_, true>), grid, block, 0, s, in, in2, weight, bias, out, C, H, W, ph, pw, dh, dw); \
        else hipLaunchKernelGGL((dwconv_kernel<KH_, KW_, false>), grid, block, 0, s, in, in2, weight, bias, out, C, H, W, ph, pw, dh, dw);  \
    } else
    UNETMI_DW(3, 3) UNETMI_DW(1, 3) UNETMI_DW(3, 1)
        return api_fail(UNETMI_E_UNSUPPORTED, "dwconv: kernel size not built (3x3, 1x3, 3x1)");
#undef UNETMI_DW
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("dwconv launch: ") + hipGetErrorString(e));
}

int unetmi_layernorm_f_fwd(int N, int C, int F, int T, int C_out, int c_off, float eps, const void* x,
                           const float* gamma, const float* beta, void* out, int dtype, void* stream) {
    if (N < 0 || C < 1 || F < 1 || T < 1 || c_off < 0 || c_off + C > C_out)
        return api_fail(UNETMI_E_BAD_DESC, "layernorm: bad sizes");
    if (N == 0) return UNETMI_OK;
    if (!x || !gamma || !beta || !out) return api_fail(UNETMI_E_BAD_ARG, "layernorm: null tensor pointer");
    if ((long long)N * C > 65535) return api_fail(UNETMI_E_UNSUPPORTED, "layernorm: N*C > 65535");
    constexpr int kRows = 72;                      // rows per wave the strip kernel holds in registers
    const bool strip = (F + 3) / 4 <= kRows && (long long)F * T * 4 < (1ll << 31);
    if (strip && dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL((layernorm_f_strip_kernel<true, kRows>), dim3((T + 63) / 64, N * C), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, gamma, beta, out, C, F, T, C_out, c_off, eps);
    else if (strip)
        hipLaunchKernelGGL((layernorm_f_strip_kernel<false, kRows>), dim3((T + 63) / 64, N * C), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, gamma, beta, out, C, F, T, C_out, c_off, eps);
    else if (dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL(layernorm_f_kernel<true>, dim3((T + 255) / 256, N * C), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, gamma, beta, out, C, F, T, C_out, c_off, eps);
    else
        hipLaunchKernelGGL(layernorm_f_kernel<false>, dim3((T + 255) / 256, N * C), dim3(256), 0, static_cast<hipStream_t>(stream),
                           x, gamma, beta, out, C, F, T, C_out, c_off, eps);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("layernorm launch: ") + hipGetErrorString(e));
}

int unetmi_instnorm_act_fwd(int N, int C, int HW, const void* z, const float* gamma, const float* beta, float eps, int act,
                            const float* act_param, int act_param_per_channel, const float* gate, void* out, int dtype,
                            void* stream) {
    if (N < 0 || C < 1 || HW < 1) return api_fail(UNETMI_E_BAD_DESC, "instnorm_act: bad sizes");
    if (act < UNETMI_ACT_NONE || act > UNETMI_ACT_GELU) return api_fail(UNETMI_E_BAD_DESC, "instnorm_act: unknown activation");
    if (N == 0) return UNETMI_OK;
    if (!z || !out) return api_fail(UNETMI_E_BAD_ARG, "instnorm_act: null tensor pointer");
    if (act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "instnorm_act: LeakyReLU / PReLU needs its slope");
    if ((long long)N * C > 0x7fffffffll) return api_fail(UNETMI_E_UNSUPPORTED, "instnorm_act: too many planes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL(instnorm_act_kernel<true>, dim3(N * C), dim3(256), 0, s, z, gamma, beta, act_param,
                           act_param_per_channel, gate, out, C, HW, eps, act);
    else if (dtype == UNETMI_DTYPE_F32)
        hipLaunchKernelGGL(instnorm_act_kernel<false>, dim3(N * C), dim3(256), 0, s, z, gamma, beta, act_param,
                           act_param_per_channel, gate, out, C, HW, eps, act);
    else return api_fail(UNETMI_E_BAD_DESC, "instnorm_act: unknown dtype");
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("instnorm_act launch: ") + hipGetErrorString(e));
}

int unetmi_layernorm_w_act_fwd(int N, int C, int H, int W, const void* z, const float* gamma, const float* beta, float eps,
                               int act, const float* act_param, int act_param_per_channel, const float* gate, void* out,
                               int dtype, void* stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1) return api_fail(UNETMI_E_BAD_DESC, "layernorm_w_act: bad sizes");
    if (act < UNETMI_ACT_NONE || act > UNETMI_ACT_GELU) return api_fail(UNETMI_E_BAD_DESC, "layernorm_w_act: unknown activation");
    if (N == 0) return UNETMI_OK;
    if (!z || !out) return api_fail(UNETMI_E_BAD_ARG, "layernorm_w_act: null tensor pointer");
    if (beta && !gamma) return api_fail(UNETMI_E_BAD_ARG, "layernorm_w_act: beta without gamma");
    if (act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "layernorm_w_act: LeakyReLU / PReLU needs its slope");
    const long long rows = (long long)N * C * H;
    if ((rows + 3) / 4 > 0x7fffffffll) return api_fail(UNETMI_E_UNSUPPORTED, "layernorm_w_act: too many rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL(rownorm_act_kernel<true>, grid, dim3(256), 0, s, z, gamma, beta, act_param, act_param_per_channel, gate,
                           out, C, H, W, rows, eps, act);
    else if (dtype == UNETMI_DTYPE_F32)
        hipLaunchKernelGGL(rownorm_act_kernel<false>, grid, dim3(256), 0, s, z, gamma, beta, act_param, act_param_per_channel, gate,
                           out, C, H, W, rows, eps, act);
    else return api_fail(UNETMI_E_BAD_DESC, "layernorm_w_act: unknown dtype");
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("layernorm_w_act launch: ") + hipGetErrorString(e));
}

int unetmi_tanh_mask_mul(int N, int K, int Cx, int F, int T, float eps, const void* y, const void* x, void* out,
                         int dtype, void* stream) {
    if (N < 0 || K < 1 || Cx < K || F < 1 || T < 1) return api_fail(UNETMI_E_BAD_DESC, "tanh_mask: bad sizes");
    if (N == 0) return UNETMI_OK;
    if (!y || !x || !out) return api_fail(UNETMI_E_BAD_ARG, "tanh_mask: null tensor pointer");
    const size_t total = (size_t)N * K * F * T;
    if (dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL(tanh_mask_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           y, x, out, K, Cx, (size_t)F * T, total, eps);
    else
        hipLaunchKernelGGL(tanh_mask_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           y, x, out, K, Cx, (size_t)F * T, total, eps);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("tanh_mask launch: ") + hipGetErrorString(e));
}

int unetmi_glu_fwd(int N, int K, int HW, const void* z, const float* scale, const float* shift, int act,
                   const float* act_param, const float* gate, void* out, int dtype, void* stream) {
    if (N < 0 || K < 1 || HW < 1) return api_fail(UNETMI_E_BAD_DESC, "glu: bad sizes");
    if (act < UNETMI_ACT_NONE || act > UNETMI_ACT_GELU) return api_fail(UNETMI_E_BAD_DESC, "glu: unknown activation");
    if (N == 0) return UNETMI_OK;
    if (!z || !out) return api_fail(UNETMI_E_BAD_ARG, "glu: null tensor pointer");
    if (act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "glu: LeakyReLU / PReLU needs its slope");
    if ((long long)N * K > 65535) return api_fail(UNETMI_E_UNSUPPORTED, "glu: N*K > 65535");
    const dim3 grid((HW + 1023) / 1024, N * K);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == UNETMI_DTYPE_BF16)
        hipLaunchKernelGGL(glu_kernel<true>, grid, dim3(256), 0, s, z, scale, shift, act_param, gate, out, K, HW, act);
    else
        hipLaunchKernelGGL(glu_kernel<false>, grid, dim3(256), 0, s, z, scale, shift, act_param, gate, out, K, HW, act);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("glu launch: ") + hipGetErrorString(e));
}

int unetmi_crf_apply(int B, int N, int C, int PF, int PT, int F, int T, const float* Hf, const float* X, float* Y,
                     void* stream) {
    if (B < 0 || N < 1 || C < 1 || PF < 1 || PT < 1 || !(PF & 1) || !(PT & 1) || F < 1 || T < 1)
        return api_fail(UNETMI_E_BAD_DESC, "crf_apply: bad sizes (tap counts must be odd)");
    if (B == 0) return UNETMI_OK;
    if (!Hf || !X || !Y) return api_fail(UNETMI_E_BAD_ARG, "crf_apply: null tensor pointer");
    const size_t total = (size_t)B * N * C * F * T;
    if ((total + 255) / 256 > 0x7fffffffull) return api_fail(UNETMI_E_UNSUPPORTED, "crf_apply: too many elements");
    hipLaunchKernelGGL(crf_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float2*>(Hf), reinterpret_cast<const float2*>(X),
                       reinterpret_cast<float2*>(Y), N, C, PF, PT, F, T, total);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("crf_apply launch: ") + hipGetErrorString(e));
}

}  // extern "C"
