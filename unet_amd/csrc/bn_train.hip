// Train-mode BatchNorm2d + activation around the convolutions, forward and backward.
// All kernels are bandwidth-bound: lanes run along the contiguous (F*T) axis, per-channel
// reductions are done in double in two stages (per-block partials in a caller-provided
// workspace, then a one-block-per-channel finish), so results do not depend on launch order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

__device__ __forceinline__ float act_value(int act, float u, float p) {
    switch (act) {
        case UNETMI_ACT_LEAKY: return u >= 0.f ? u : p * u;
        case UNETMI_ACT_RELU: return u > 0.f ? u : 0.f;
        case UNETMI_ACT_SILU: return u * fast_sigmoid(u);
        case UNETMI_ACT_SOFTPLUS: return fast_softplus(u);
        case UNETMI_ACT_SIGMOID: return fast_sigmoid(u);
        case UNETMI_ACT_TANH: return fast_tanh(u);
        case UNETMI_ACT_GELU: return 0.5f * u * (1.f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}

// d act(u) / du
__device__ __forceinline__ float act_slope(int act, float u, float p) {
    switch (act) {
        case UNETMI_ACT_LEAKY: return u > 0.f ? 1.f : p;
        case UNETMI_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case UNETMI_ACT_SILU: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f + u * (1.f - s)); }
        case UNETMI_ACT_SOFTPLUS: return u > 20.f ? 1.f : 1.f / (1.f + expf(-u));
        case UNETMI_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-u)); return s * (1.f - s); }
        case UNETMI_ACT_TANH: { const float t = tanhf(u); return 1.f - t * t; }
        case UNETMI_ACT_GELU:
            return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.39894228040143268f * expf(-0.5f * u * u);
        default: return 1.f;
    }
}

struct BnArgs {
    const float* z;        // conv output incl. bias [N,C,HW]
    const float* dy;       // gradient w.r.t. the activation output (backward only)
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* act_param;
    float* out;            // y (forward) or dz (backward)
    double* ws;            // [C][S][2] partials, then (backward) [C][S] for the bias gradient
    int N, C, HW, S, act;
    int batch_stats;       // backward: 1 = mean/invstd are statistics of this batch, 0 = constants
    long long chunk;       // elements of one channel handled by one block
    long long dy_bs;       // batch stride of dy in elements
};

__device__ __forceinline__ void block_sum2(double& a, double& b) {
    __shared__ double sa[4], sb[4];
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) { a = sa[0] + sa[1] + sa[2] + sa[3]; b = sb[0] + sb[1] + sb[2] + sb[3]; }
    __syncthreads();
}

// MODE 0: (sum z, sum z^2).  MODE 1: g = dy * act'(u): (sum g, sum g * xhat).
// grid (S, C): block s of channel c walks elements [s*chunk, (s+1)*chunk) of the N*HW values of c.
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(BnArgs a) {
    const int c = blockIdx.y, s = blockIdx.x;
    const long long M = (long long)a.N * a.HW;
    const long long lo = (long long)s * a.chunk, hi = lo + a.chunk < M ? lo + a.chunk : M;
    double s0 = 0.0, s1 = 0.0;
    float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f, ap = 0.f;
    if (MODE == 1) {
        mu = a.mean[c]; is = a.invstd[c]; ga = a.gamma[c]; be = a.beta[c];
        ap = a.act_param ? a.act_param[0] : 0.f;
    }
    long long i = lo + threadIdx.x;
    if (i < hi) {
        long long n = i / a.HW;
        int hw = (int)(i - n * a.HW);
        size_t off = ((size_t)n * a.C + c) * a.HW + hw;
        const size_t jump = (size_t)(a.C - 1) * a.HW;     // extra offset when stepping into the next sample
        size_t offd = MODE == 1 ? (size_t)n * a.dy_bs + (size_t)c * a.HW + hw : 0;
        const size_t jumpd = MODE == 1 ? (size_t)(a.dy_bs - a.HW) : 0;
        auto step = [&]() {                                 // this thread's next element: 256 further on
            hw += 256; off += 256; offd += 256;
            while (hw >= a.HW) { hw -= a.HW; off += jump; offd += jumpd; }
        };
        auto add = [&](float z, float d) {
            if (MODE == 0) {
                s0 += z;
                s1 += (double)z * z;
            } else {
                const float xh = (z - mu) * is;
                const float g = d * act_slope(a.act, fmaf(xh, ga, be), ap);
                s0 += g;
                s1 += (double)g * xh;
            }
        };
        // four elements per pass, their loads issued together: one 4-byte load in flight per thread made this a
        // chain of memory latencies (3 TB/s); the sums are still added in element order
        for (; i + 3 * 256 < hi; i += 4 * 256) {
            float z[4], d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                z[q] = a.z[off];
                if (MODE == 1) d[q] = a.dy[offd];
                step();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) add(z[q], d[q]);
        }
        for (; i < hi; i += 256) {
            const float z = a.z[off];
            add(z, MODE == 1 ? a.dy[offd] : 0.f);
            step();
        }
    }
    block_sum2(s0, s1);
    if (threadIdx.x == 0) {
        a.ws[((size_t)c * a.S + s) * 2 + 0] = s0;
        a.ws[((size_t)c * a.S + s) * 2 + 1] = s1;
    }
}

// One block per channel: batch mean, 1/sqrt(biased var + eps), running-stat update with the
// unbiased variance (torch.nn.BatchNorm2d train-mode semantics).
__global__ __launch_bounds__(64) void bn_stats_finish_kernel(const double* ws, int S, long long M, float eps,
                                                             float momentum, float* running_mean, float* running_var,
                                                             float* mean, float* invstd) {
    const int c = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int s = threadIdx.x; s < S; s += 64) { s0 += ws[((size_t)c * S + s) * 2]; s1 += ws[((size_t)c * S + s) * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    if (threadIdx.x == 0) {
        const double m = s0 / (double)M;
        double var = s1 / (double)M - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        if (running_var) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

__global__ __launch_bounds__(64) void bn_bwd_finish_kernel(const double* ws, int S, float* dgamma, float* dbeta,
                                                           double* sums /* [C][2] */) {
    const int c = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int s = threadIdx.x; s < S; s += 64) { s0 += ws[((size_t)c * S + s) * 2]; s1 += ws[((size_t)c * S + s) * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    if (threadIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)s0;
        if (dgamma) dgamma[c] = (float)s1;
        sums[c * 2] = s0;
        sums[c * 2 + 1] = s1;
    }
}

__global__ __launch_bounds__(64) void chan_sum_finish_kernel(const double* ws, int S, float* out) {
    const int c = blockIdx.x;
    double s0 = 0.0;
    for (int s = threadIdx.x; s < S; s += 64) s0 += ws[(size_t)c * S + s];
    for (int o = 32; o > 0; o >>= 1) s0 += __shfl_down(s0, o, 64);
    if (threadIdx.x == 0) out[c] = (float)s0;
}

// y = act((z - mean) * invstd * gamma + beta); grid (ceil(HW/1024), N*C)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnArgs a) {
    const int nc = blockIdx.y, c = nc % a.C;
    const float is = a.invstd[c];
    const float ap = a.act_param ? a.act_param[0] : 0.f;
    const float* src = a.z + (size_t)nc * a.HW;
    float* dst = a.out + (size_t)nc * a.HW;
    const int base = blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 256;
        // same operation order as (z - mean) * invstd * gamma + beta to stay within rounding of it
        if (i < a.HW) dst[i] = act_value(a.act, fmaf((src[i] - a.mean[c]) * is, a.gamma[c], a.beta[c]), ap);
    }
}

// dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * act'(u); also per-block
// partial sums of dz (the conv-bias gradient).  grid (S, C) like the reductions.
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BnArgs a, const double* sums, double* ws_bias) {
    const int c = blockIdx.y, s = blockIdx.x;
    const long long M = (long long)a.N * a.HW;
    const long long lo = (long long)s * a.chunk, hi = lo + a.chunk < M ? lo + a.chunk : M;
    const float mu = a.mean[c], is = a.invstd[c], ga = a.gamma[c], be = a.beta[c];
    const float ap = a.act_param ? a.act_param[0] : 0.f;
    const float mg = a.batch_stats ? (float)(sums[c * 2] / (double)M) : 0.f;
    const float mgx = a.batch_stats ? (float)(sums[c * 2 + 1] / (double)M) : 0.f;
    const float k = ga * is;
    double sb = 0.0, unused = 0.0;
    long long i = lo + threadIdx.x;
    if (i < hi) {
        long long n = i / a.HW;
        int hw = (int)(i - n * a.HW);
        size_t off = ((size_t)n * a.C + c) * a.HW + hw;
        const size_t jump = (size_t)(a.C - 1) * a.HW;
        size_t offd = (size_t)n * a.dy_bs + (size_t)c * a.HW + hw;
        const size_t jumpd = (size_t)(a.dy_bs - a.HW);
        for (; i < hi; i += 256) {
            const float xh = (a.z[off] - mu) * is;
            const float g = a.dy[offd] * act_slope(a.act, fmaf(xh, ga, be), ap);
            const float dz = k * (g - mg - xh * mgx);
            a.out[off] = dz;
            sb += dz;
            hw += 256; off += 256; offd += 256;
            while (hw >= a.HW) { hw -= a.HW; off += jump; offd += jumpd; }
        }
    }
    block_sum2(sb, unused);
    if (threadIdx.x == 0) ws_bias[(size_t)c * a.S + s] = sb;
}

// torch.optim.Adam (no amsgrad, L2 weight decay added to the gradient), one fused pass over a flat bucket
__global__ __launch_bounds__(256) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                   float b2, float eps, float wd, float gscale, float bc1, float sqrt_bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float grad = g[i] * gscale;
    const float w = p[i];
    if (wd != 0.f) grad = fmaf(wd, w, grad);
    const float mi = m[i] + (grad - m[i]) * (1.f - b1);
    const float vi = b2 * v[i] + (1.f - b2) * grad * grad;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = w - (lr / bc1) * (mi / denom);
}

static int splits_for(int N, int C, int HW, long long* chunk) {
    const long long M = (long long)N * HW;
    long long S = (4096 + C - 1) / C;                 // aim for >= 4096 blocks (256 CUs x 16)
    const long long maxS = (M + 2047) / 2048;         // at least 8 elements per thread
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    long long ch = (M + S - 1) / S;
    ch = (ch + 255) / 256 * 256;
    S = (M + ch - 1) / ch;
    *chunk = ch;
    return (int)S;
}

static size_t ws_bytes_for(int N, int C, int HW) {
    long long chunk;
    const int S = splits_for(N, C, HW, &chunk);
    // [C][S][2] partials + [C][2] sums + [C][S] bias partials
    return ((size_t)C * S * 3 + (size_t)C * 2) * sizeof(double);
}

static int check_sizes(const char* what, int N, int C, int HW) {
    if (N < 0 || C < 1 || HW < 1) return api_fail(UNETMI_E_BAD_DESC, std::string(what) + ": bad sizes");
    if (C > 65535) return api_fail(UNETMI_E_UNSUPPORTED, std::string(what) + ": more than 65535 channels");
    if ((long long)N * C > 65535) return api_fail(UNETMI_E_UNSUPPORTED, std::string(what) + ": N*C > 65535");
    return UNETMI_OK;
}

// Scratch of the entry point that must not allocate: the per-block partial sums live in this file-scope device
// array (every GPU the library runs on holds its own copy until the process ends).  In host code the array's name
// is only a stand-in for the symbol; its device address is looked up once per device and kept.
constexpr size_t kStaticScratchDoubles = 32768;                   // 256 KiB: C * splits * 2 partial sums
static __device__ __align__(256) double scratch[kStaticScratchDoubles];
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static double* g_scratch_ptr[kMaxDevices];                         // guarded by g_scratch_mutex

// The array's copy on the device that owns `stream`, which must be the calling thread's current device (the lookup
// resolves the symbol on the current device; handing a kernel on one GPU another GPU's address would fault).
static int static_scratch_of_stream_device(hipStream_t stream, double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("static scratch: hipGetDevice: ") + hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return api_fail(UNETMI_E_UNSUPPORTED, "static scratch: device index out of range");
    if (stream) {                                  // the null stream belongs to the current device by definition
        hipDevice_t sdev = -1;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess)
            return api_fail(UNETMI_E_HIP, std::string("static scratch: hipStreamGetDevice: ") + hipGetErrorString(e));
        if ((int)sdev != dev)
            return api_fail(UNETMI_E_BAD_ARG, "static scratch: the stream belongs to device " + std::to_string((int)sdev) +
                                                  " but device " + std::to_string(dev) + " is current");
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return api_fail(UNETMI_E_HIP, std::string("static scratch: hipGetSymbolAddress: ") + hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return UNETMI_OK;
}

static int launched(const char* what) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
}

// running statistics of train-mode BatchNorm from a batch's (mean, invstd): var_biased = 1/invstd^2 - eps,
// running_var takes the unbiased one (count / (count - 1)), as torch.nn.BatchNorm2d does
__global__ __launch_bounds__(256) void bn_running_update_kernel(int C, double count, float eps, float momentum,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double is = (double)invstd[c];
    double var = 1.0 / (is * is) - (double)eps;
    if (var < 0.0) var = 0.0;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
}

// packed[i] = map[i] >= 0 ? weight[map[i]] : 0 -- re-packs a weight tensor into a kernel's image on the device
// (the host-side pack is a permutation with zero padding, so it is one gather once the map is known).
__global__ __launch_bounds__(256) void gather_kernel(size_t n, const float* __restrict__ src,
                                                     const int* __restrict__ map, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int m = map[i];
    dst[i] = m >= 0 ? src[m] : 0.f;
}

// mean / invstd from per-channel partials [C][S][2] that another kernel left (the convolution's statistics epilogue)
int launch_bn_stats_finish(const double* ws, int C, int S, long long M, float eps, float* mean, float* invstd, hipStream_t s) {
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(C), dim3(64), 0, s, ws, S, M, eps, 0.f, static_cast<float*>(nullptr),
                       static_cast<float*>(nullptr), mean, invstd);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("bn_stats_finish launch: ") + hipGetErrorString(e));
}

}  // namespace unetmi

using namespace unetmi;

extern "C" {

size_t unetmi_bn_workspace_bytes(int N, int C, int HW) {
    if (N < 1 || C < 1 || HW < 1) return 0;
    return ws_bytes_for(N, C, HW);
}

int unetmi_bn_train_stats(int N, int C, int HW, const float* z, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_sizes("bn_train_stats", N, C, HW)) return rc;
    if (N == 0) return api_fail(UNETMI_E_BAD_ARG, "bn_train_stats: batch statistics of an empty batch");
    if (!z || !mean || !invstd || !ws) return api_fail(UNETMI_E_BAD_ARG, "bn_train_stats: null pointer");
    if (ws_bytes < ws_bytes_for(N, C, HW)) return api_fail(UNETMI_E_BAD_ARG, "bn_train_stats: workspace too small");
    BnArgs a{};
    a.z = z; a.N = N; a.C = C; a.HW = HW; a.ws = static_cast<double*>(ws);
    a.S = splits_for(N, C, HW, &a.chunk);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(a.S, C), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(C), dim3(64), 0, s, a.ws, a.S, (long long)N * HW, eps, momentum,
                       running_mean, running_var, mean, invstd);
    return launched("bn_train_stats");
}

int unetmi_bn_train_stats_static(int N, int C, int HW, const float* z, float eps, float momentum,
                                 float* running_mean, float* running_var, float* mean, float* invstd, void* stream) {
    if (int rc = check_sizes("bn_train_stats_static", N, C, HW)) return rc;
    if (N == 0) return api_fail(UNETMI_E_BAD_ARG, "bn_train_stats_static: batch statistics of an empty batch");
    if (!z || !mean || !invstd) return api_fail(UNETMI_E_BAD_ARG, "bn_train_stats_static: null pointer");
    BnArgs a{};
    a.z = z; a.N = N; a.C = C; a.HW = HW;
    a.S = splits_for(N, C, HW, &a.chunk);
    if ((size_t)C * a.S * 2 > kStaticScratchDoubles)
        return api_fail(UNETMI_E_UNSUPPORTED,
                        "bn_train_stats_static: " + std::to_string((size_t)C * a.S * 2 * sizeof(double)) +
                            " bytes of partial sums exceed the " + std::to_string(kStaticScratchDoubles * sizeof(double)) +
                            "-byte static scratch; use unetmi_bn_train_stats with a workspace");
    double* ws = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = static_scratch_of_stream_device(s, &ws)) return rc;
    a.ws = ws;
    // every partial the finish kernel reads is written by the reduce kernel of this call: nothing to clear
    hipLaunchKernelGGL(bn_reduce_kernel<0>, dim3(a.S, C), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_stats_finish_kernel, dim3(C), dim3(64), 0, s, a.ws, a.S, (long long)N * HW, eps, momentum,
                       running_mean, running_var, mean, invstd);
    return launched("bn_train_stats_static");
}

int unetmi_bn_act_fwd(int N, int C, int HW, const float* z, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, const float* act_param, float* out,
                      void* stream) {
    if (int rc = check_sizes("bn_act_fwd", N, C, HW)) return rc;
    if (N == 0) return UNETMI_OK;
    if (!z || !mean || !invstd || !gamma || !beta || !out) return api_fail(UNETMI_E_BAD_ARG, "bn_act_fwd: null pointer");
    if (act < UNETMI_ACT_NONE || act > UNETMI_ACT_GELU) return api_fail(UNETMI_E_BAD_DESC, "bn_act_fwd: unknown activation");
    if (act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "bn_act_fwd: LeakyReLU needs its slope");
    BnArgs a{};
    a.z = z; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.act = act; a.act_param = act_param;
    a.out = out; a.N = N; a.C = C; a.HW = HW;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3((HW + 1023) / 1024, N * C), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return launched("bn_act_fwd");
}

int unetmi_bn_act_bwd(int N, int C, int HW, const float* dy, long long dy_batch_stride, const float* z,
                      const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                      const float* act_param, int batch_stats, float* dz, float* dgamma, float* dbeta, float* dbias,
                      void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_sizes("bn_act_bwd", N, C, HW)) return rc;
    if (N == 0) return api_fail(UNETMI_E_BAD_ARG, "bn_act_bwd: empty batch");
    if (!dy || !z || !mean || !invstd || !gamma || !beta || !dz || !ws)
        return api_fail(UNETMI_E_BAD_ARG, "bn_act_bwd: null pointer");
    if (act < UNETMI_ACT_NONE || act > UNETMI_ACT_GELU) return api_fail(UNETMI_E_BAD_DESC, "bn_act_bwd: unknown activation");
    if (act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "bn_act_bwd: LeakyReLU needs its slope");
    if (ws_bytes < ws_bytes_for(N, C, HW)) return api_fail(UNETMI_E_BAD_ARG, "bn_act_bwd: workspace too small");
    if (dy_batch_stride < (long long)C * HW) return api_fail(UNETMI_E_BAD_ARG, "bn_act_bwd: dy batch stride smaller than C*HW");
    BnArgs a{};
    a.dy_bs = dy_batch_stride;
    a.batch_stats = batch_stats ? 1 : 0;
    a.z = z; a.dy = dy; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.act = act;
    a.act_param = act_param; a.out = dz; a.N = N; a.C = C; a.HW = HW; a.ws = static_cast<double*>(ws);
    a.S = splits_for(N, C, HW, &a.chunk);
    double* sums = a.ws + (size_t)C * a.S * 2;
    double* ws_bias = sums + (size_t)C * 2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_reduce_kernel<1>, dim3(a.S, C), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(64), 0, s, a.ws, a.S, dgamma, dbeta, sums);
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(a.S, C), dim3(256), 0, s, a, sums, ws_bias);
    if (dbias) hipLaunchKernelGGL(chan_sum_finish_kernel, dim3(C), dim3(64), 0, s, ws_bias, a.S, dbias);
    return launched("bn_act_bwd");
}

int unetmi_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step, void* stream) {
    if (step < 1) return api_fail(UNETMI_E_BAD_ARG, "adam_step: step counts from 1");
    if (n == 0) return UNETMI_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return api_fail(UNETMI_E_BAD_ARG, "adam_step: null pointer");
    if ((n + 255) / 256 > 0x7fffffffull) return api_fail(UNETMI_E_UNSUPPORTED, "adam_step: bucket too large");
    const float bc1 = (float)(1.0 - pow((double)beta1, step));
    const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)beta2, step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                       param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, bc1, sqrt_bc2);
    return launched("adam_step");
}

int unetmi_gather_f32(size_t n, const float* src, const int* map, float* dst, void* stream) {
    if (n == 0) return UNETMI_OK;
    if (!src || !map || !dst) return api_fail(UNETMI_E_BAD_ARG, "gather_f32: null pointer");
    if ((n + 255) / 256 > 0x7fffffffull) return api_fail(UNETMI_E_UNSUPPORTED, "gather_f32: too many elements");
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), n,
                       src, map, dst);
    return launched("gather_f32");
}

int unetmi_bn_running_update(int C, long long count, float eps, float momentum, const float* mean, const float* invstd,
                             float* running_mean, float* running_var, void* stream) {
    if (C < 1 || count < 1) return api_fail(UNETMI_E_BAD_DESC, "bn_running_update: bad sizes");
    if (!mean || !invstd || !running_mean || !running_var) return api_fail(UNETMI_E_BAD_ARG, "bn_running_update: null pointer");
    hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), C,
                       (double)count, eps, momentum, mean, invstd, running_mean, running_var);
    return launched("bn_running_update");
}

}  // extern "C"
