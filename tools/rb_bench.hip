// Standalone timing harness for the residual-block layer kernel (resblock_dma.hip is compiled into this binary,
// optionally with -DRB_ABLATE=<bits> to switch stages off and price them).  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DRB_ABLATE=n] tools/rb_bench.hip -o tools/rb_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../unet_amd/csrc/resblock_dma.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 257, W = argc > 3 ? atoi(argv[3]) : 501;
    const bool bf16 = argc > 4 && atoi(argv[4]) != 0;      // the buffers are oversized for bf16; values are arbitrary bits
    const size_t n = (size_t)N * 64 * H * W;
    std::vector<float> hx(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    float *x, *out, *par;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&par, 8192 * 4));
    CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> hp(8192);
    for (int i = 0; i < 8192; ++i) { s = s * 1664525u + 1013904223u; hp[i] = (((s >> 8) & 0xffff) / 32768.f - 1.f) * 0.2f; }
    CK(hipMemcpy(par, hp.data(), 8192 * 4, hipMemcpyHostToDevice));
    unetmi::ResblockArgs a{};
    a.x = x; a.out = out; a.dw_w = par; a.dw_b = par + 600; a.pw_w = par + 1024; a.pw_b = par + 5200; a.alpha = par + 5300;
    a.scale = par + 5400; a.shift = par + 5500; a.N = N; a.H = H; a.W = W; a.residual = 1; a.act_first = 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ds[6] = {16, 1, 2, 4, 8, 16};
    for (int di = 0; di < 6; ++di) {
        for (int i = 0; i < 2; ++i) CK(unetmi::launch_resblock_dma(ds[di], nullptr, a, bf16));
        CK(hipEventRecord(e0, nullptr));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) CK(unetmi::launch_resblock_dma(ds[di], nullptr, a, bf16));
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double pix = (double)N * H * W;
        printf("d=%-2d %.3f ms  %.2f TB/s algorithmic  %.1f ps/pixel\n", ds[di], ms, 2.0 * n * (bf16 ? 2 : 4) / ms / 1e9, ms * 1e9 / pix);
    }
    return 0;
}
