// Weight and bias gradient of the depthwise 3x3 / 1x3 / 3x1 'same' convolution of a residual-block layer
// (UNet_m.py:546-548) for train mode:
//     dW[c][r][s] = sum_{n,y,x} dY[n,c,y,x] * X[n,c, y - ph + r*dh, x - pw + s*dw],      db[c] = sum dY[n,c,y,x]
// A pure reduction per channel: block (split, c) walks its share of the (n, y) rows with lanes along time, keeps
// kh*kw + 1 partial sums per thread, folds them across the block and writes one partial row; a second kernel adds
// the splits in a fixed order (no atomics: bit-reproducible).  The input gradient needs no kernel of its own: it is
// the forward kernel applied to dY with the taps reversed.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/unetmi.h"

namespace unetmi {

int api_fail(int code, const std::string& msg);

constexpr int kDwSplits = 32;

template <int KH, int KW>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ x2,
                                                       const float* __restrict__ dy, float* __restrict__ partial, int N,
                                                       int C, int H, int W, int ph, int pw, int dh, int dw) {
    constexpr int NT = KH * KW;
    const int split = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    float acc[NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) acc[i] = 0.f;
    for (int row = split; row < N * H; row += kDwSplits) {
        const int n = row / H, y = row - n * H;
        const size_t plane = ((size_t)n * C + c) * H * W;
        for (int col = tid; col < W; col += 256) {
            const float g = dy[plane + (size_t)y * W + col];
            acc[NT] += g;
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                const int iy = y - ph + r * dh;
                if ((unsigned)iy >= (unsigned)H) continue;           // uniform over the block
#pragma unroll
                for (int s = 0; s < KW; ++s) {
                    const int ix = col - pw + s * dw;
                    float v = 0.f;
                    if ((unsigned)ix < (unsigned)W) {
                        v = x[plane + (size_t)iy * W + ix];
                        if (x2) v += x2[plane + (size_t)iy * W + ix];
                    }
                    acc[r * KW + s] = fmaf(g, v, acc[r * KW + s]);
                }
            }
        }
    }
    __shared__ float red[4][NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid <= NT) partial[((size_t)c * kDwSplits + split) * (NT + 1) + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(64) void dw_wgrad_finish_kernel(const float* __restrict__ partial, float* __restrict__ dweight,
                                                             float* __restrict__ dbias, int C, int NT) {
    const int i = blockIdx.x * 64 + threadIdx.x;          // c * (NT+1) + tap
    if (i >= C * (NT + 1)) return;
    const int c = i / (NT + 1), tap = i - c * (NT + 1);
    double s = 0.0;
    for (int k = 0; k < kDwSplits; ++k) s += (double)partial[((size_t)c * kDwSplits + k) * (NT + 1) + tap];
    if (tap < NT) dweight[c * NT + tap] = (float)s;
    else if (dbias) dbias[c] = (float)s;
}

}  // namespace unetmi

using namespace unetmi;

extern "C" {

size_t unetmi_dwconv2d_wgrad_workspace_bytes(int C, int kh, int kw) {
    if (C < 1 || kh < 1 || kw < 1) return 0;
    return (size_t)C * kDwSplits * (kh * kw + 1) * sizeof(float);
}

int unetmi_dwconv2d_wgrad(int N, int C, int H, int W, int kh, int kw, int ph, int pw, int dh, int dw, const float* in,
                          const float* in2, const float* dout, float* dweight, float* dbias, void* ws, size_t ws_bytes,
                          void* stream) {
    if (N < 0 || C < 1 || H < 1 || W < 1) return api_fail(UNETMI_E_BAD_DESC, "dwconv2d_wgrad: bad sizes");
    if (2 * ph != dh * (kh - 1) || 2 * pw != dw * (kw - 1))
        return api_fail(UNETMI_E_UNSUPPORTED, "dwconv2d_wgrad: only 'same' padding (2p = d(k-1)) is built");
    if (!dweight) return api_fail(UNETMI_E_BAD_ARG, "dwconv2d_wgrad: null gradient pointer");
    if (C > 65535) return api_fail(UNETMI_E_UNSUPPORTED, "dwconv2d_wgrad: more than 65535 channels");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int NT = kh * kw;
    if (!ws || (N > 0 && (!in || !dout))) return api_fail(UNETMI_E_BAD_ARG, "dwconv2d_wgrad: null tensor pointer");
    if (ws_bytes < unetmi_dwconv2d_wgrad_workspace_bytes(C, kh, kw)) return api_fail(UNETMI_E_BAD_ARG, "dwconv2d_wgrad: workspace too small");
    float* partial = static_cast<float*>(ws);
    const dim3 grid(kDwSplits, C), block(256);
    if (kh == 3 && kw == 3) hipLaunchKernelGGL((dw_wgrad_kernel<3, 3>), grid, block, 0, s, in, in2, dout, partial, N, C, H, W, ph, pw, dh, dw);
    else if (kh == 1 && kw == 3) hipLaunchKernelGGL((dw_wgrad_kernel<1, 3>), grid, block, 0, s, in, in2, dout, partial, N, C, H, W, ph, pw, dh, dw);
    else if (kh == 3 && kw == 1) hipLaunchKernelGGL((dw_wgrad_kernel<3, 1>), grid, block, 0, s, in, in2, dout, partial, N, C, H, W, ph, pw, dh, dw);
    else return api_fail(UNETMI_E_UNSUPPORTED, "dwconv2d_wgrad: kernel size not built (3x3, 1x3, 3x1)");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("dwconv2d_wgrad launch: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(dw_wgrad_finish_kernel, dim3((C * (NT + 1) + 63) / 64), dim3(64), 0, s, partial, dweight, dbias, C, NT);
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("dwconv2d_wgrad finish launch: ") + hipGetErrorString(e));
}

}  // extern "C"
