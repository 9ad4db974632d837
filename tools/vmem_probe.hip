// Developer probe: how long a wave takes to ISSUE (and to complete) a burst of 32
// global loads / stores when the other wave on its SIMD is running back-to-back
// fp32 MFMAs, versus when that wave is idle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t;
}

template <bool MFMA_BESIDE, bool STORES, int PRIO>
__global__ __launch_bounds__(512, 2) void probe(const float* in, float* out, unsigned long long* res, int reps) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave < 4) {
        if (!MFMA_BESIDE) return;
        f32x16 acc[4];
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float a = 1.f + tid * 1e-3f, b = 0.5f;
        for (int k = 0; k < reps * 260; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f; for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[(size_t)blockIdx.x * 512 + tid] = s;
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    unsigned long long t_issue = 0, t_done = 0;
    const size_t base = ((size_t)blockIdx.x * 4 + (wave - 4)) * 32 * 1024 * 16;   // floats; private 2 MB per wave
    for (int rep = 0; rep < reps; ++rep) {
        float v[32];
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t0 = now();
        __builtin_amdgcn_sched_barrier(0);
        if (STORES) {
#pragma unroll
            for (int i = 0; i < 32; ++i) out[base + (size_t)(rep & 15) * 32 * 1024 + i * 1024 + lane] = (float)i;
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = in[base + (size_t)(rep & 15) * 32 * 1024 + i * 1024 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = now();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = now();
        if (!STORES) { float s = 0.f; for (int i = 0; i < 32; ++i) s += v[i]; if (s == 123.25f) out[0] = s; }
        t_issue += t1 - t0; t_done += t2 - t0;
    }
    if (lane == 0) { atomicAdd(&res[0], t_issue); atomicAdd(&res[1], t_done); atomicAdd(&res[2], (unsigned long long)reps); }
}

template <bool M, bool S, int P> void run(const char* name, const float* in, float* out, unsigned long long* res) {
    CK(hipMemset(res, 0, 24));
    hipLaunchKernelGGL((probe<M, S, P>), dim3(256), dim3(512), 0, 0, in, out, res, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[3]; CK(hipMemcpy(h, res, 24, hipMemcpyDeviceToHost));
    printf("%-44s issue of 32: %8.0f cycles   until complete: %8.0f cycles\n", name, (double)h[0] / h[2], (double)h[1] / h[2]);
}

int main() {
    const size_t n = (size_t)256 * 4 * 32 * 1024 * 16 + 1024;
    float *in, *out; unsigned long long* res;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&res, 24));
    CK(hipMemset(in, 0, n * 4));
    run<false, false, 0>("loads, partner wave idle", in, out, res);
    run<true, false, 0>("loads, partner wave in MFMA loop", in, out, res);
    run<true, false, 3>("loads, partner in MFMA loop, s_setprio 3", in, out, res);
    run<false, true, 0>("stores, partner wave idle", in, out, res);
    run<true, true, 0>("stores, partner wave in MFMA loop", in, out, res);
    run<true, true, 3>("stores, partner in MFMA loop, s_setprio 3", in, out, res);
    return 0;
}
