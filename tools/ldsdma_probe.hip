// Developer probe: semantics of buffer_load ... lds (LDS-DMA) on gfx950 -- where each lane's
// value lands and what out-of-range lanes write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const float* in, float* out, int nvalid) {
    __shared__ float lds[64 * 4 * 2 + 64];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 2 + 64; i += 64) lds[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nvalid * 4, 0x00020000);
    // dword: lane l reads element (63 - l) * 3 (arbitrary per-lane source); lanes whose offset >= nvalid*4 are out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + 8), 4, (63 - lane) * 3 * 4, 0, 0, 0);
    // 16 bytes per lane
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + 128), 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 * 2 + 64; i += 64) out[i] = lds[i];
}

int main() {
    float *in, *out; const int n = 1024;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, 4096));
    float h[1024]; for (int i = 0; i < n; ++i) h[i] = (float)i;
    CK(hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, out, 100);
    CK(hipDeviceSynchronize());
    float r[576]; CK(hipMemcpy(r, out, 576 * 4, hipMemcpyDeviceToHost));
    printf("dword DMA (dest lds+8, lane l <- in[(63-l)*3], valid < 100):\n");
    for (int i = 0; i < 80; ++i) printf("%g ", r[i]); printf("\n");
    printf("x4 DMA (dest lds+128, lane l <- in[4l..4l+3], valid < 100):\n");
    for (int i = 120; i < 120 + 130; ++i) printf("%g ", r[i]); printf("\n");
    return 0;
}
