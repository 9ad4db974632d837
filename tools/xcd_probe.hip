// Probe: (1) which XCD a workgroup runs on (hardware register XCC_ID) against blockIdx.x % 8 for a 256-workgroup,
// one-per-CU launch; (2) the lane layout of v_mfma_f32_4x4x1_16b_f32.
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_probe.hip -o tools/xcd_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void where_kernel(int* out) {
    extern __shared__ float big[];      // forces one workgroup per CU
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) { out[blockIdx.x] = (int)(x & 0xf); big[0] = 1.f; }
}

__global__ void mfma_kernel(const float* a, const float* b, float* d) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[threadIdx.x * 4 + i] = c[i];
}

int main() {
    int* dout; hipMalloc(&dout, 256 * 4);
    hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(where_kernel, dim3(256), dim3(256), 100 * 1024, 0, dout);
        std::vector<int> h(256);
        hipMemcpy(h.data(), dout, 256 * 4, hipMemcpyDeviceToHost);
        int hist[16] = {0}, match = 0;
        for (int i = 0; i < 256; ++i) { hist[h[i] & 15]++; match += (h[i] == i % 8); }
        printf("launch %d: XCC histogram", rep);
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("  | blockIdx%%8 == XCC_ID for %d of 256\n", match);
    }
    float *a, *b, *d; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    std::vector<float> ha(64), hb(64), hd(256);
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 + l; }     // a: lane value, b: lane value
    hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
    // hypothesis: lane l = 4*blk + j holds D_blk[i][j] in c[i], D_blk[i][j] = a[4*blk + i] * b[4*blk + j]
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const int blk = l / 4, j = l % 4;
        const float want = ha[4 * blk + i] * hb[4 * blk + j];
        if (hd[l * 4 + i] != want) { if (ok) printf("mismatch lane %d reg %d: got %g want %g\n", l, i, hd[l * 4 + i], want); ok = 0; }
    }
    printf("mfma_f32_4x4x1 layout hypothesis (lane = 4*block + col, reg = row; a lane = 4*block + row): %s\n", ok ? "CONFIRMED" : "WRONG");
    if (!ok) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3]);
    return 0;
}
