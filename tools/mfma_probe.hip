// Developer probe: what a wave's fp32 MFMA loop sustains on gfx950 as LDS reads and
// scalar/vector bookkeeping are added around the matrix instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, int stride) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = 1.0f + tid * 1e-3f, b[4] = {0.5f, 0.25f, 0.125f, 1.f};
    if (MODE == 0) {            // bare MFMAs, operands in registers
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        }
    } else if (MODE == 1) {     // operands read from LDS one step ahead (5 ds_read_b32 per 4 MFMA)
        int off = tid & 63;
        float an = lds[off], bn[4];
        for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
        for (int k = 0; k < iters; ++k) {
            a = an; for (int t = 0; t < 4; ++t) b[t] = bn[t];
            off = (off + stride) & 4095;
            an = lds[off];
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        }
    } else if (MODE == 3 || MODE == 4) {  // loads of step k+1 pinned BEFORE the MFMAs of step k
        int off = tid & 63;
        float an = lds[off], bn[4];
        for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
        int tw = 0, th = 0;
        for (int k = 0; k < iters; ++k) {
            a = an; for (int t = 0; t < 4; ++t) b[t] = bn[t];
            if (MODE == 4) {   // the scalar tap walk of the conv kernel
                ++tw; off += 1;
                if (tw == 5) { tw = 0; off += stride; ++th; if (th == 7) { th = 0; off += 3; } }
                off &= 4095;
            } else {
                off = (off + stride) & 4095;
            }
            an = lds[off];
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 5) {     // register copies only (no LDS): 5 v_mov of the operands per 4 MFMA
        float an = a * 0.5f, bn[4] = {b[1], b[2], b[3], b[0]};
        for (int k = 0; k < iters; ++k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float ta = a; a = an; an = ta;
#pragma unroll
            for (int t = 0; t < 4; ++t) { float tb = b[t]; b[t] = bn[t]; bn[t] = tb; }
            asm volatile("" : "+v"(a), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 6) {     // ping-pong operand sets, unrolled by 2: no register copies
        int off = tid & 63;
        float a0 = lds[off], b0[4], a1, b1[4];
        for (int t = 0; t < 4; ++t) b0[t] = lds[off + 64 * (t + 1)];
        for (int k = 0; k < iters; k += 2) {
            off = (off + stride) & 4095;
            a1 = lds[off];
#pragma unroll
            for (int t = 0; t < 4; ++t) b1[t] = lds[off + 64 * (t + 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            off = (off + stride) & 4095;
            a0 = lds[off];
#pragma unroll
            for (int t = 0; t < 4; ++t) b0[t] = lds[off + 64 * (t + 1)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (MODE == 7) {     // ping-pong, loads issued BETWEEN the MFMAs of the other set
        int off = tid & 63;
        float a0 = lds[off], b0[4], a1, b1[4];
        for (int t = 0; t < 4; ++t) b0[t] = lds[off + 64 * (t + 1)];
        for (int k = 0; k < iters; k += 2) {
            off = (off + stride) & 4095;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[0], acc[0], 0, 0, 0);
            a1 = lds[off]; b1[0] = lds[off + 64];
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[1], acc[1], 0, 0, 0);
            b1[1] = lds[off + 128]; b1[2] = lds[off + 192];
            __builtin_amdgcn_sched_barrier(0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[2], acc[2], 0, 0, 0);
            b1[3] = lds[off + 256];
            __builtin_amdgcn_sched_barrier(0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[3], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            off = (off + stride) & 4095;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[0], acc[0], 0, 0, 0);
            a0 = lds[off]; b0[0] = lds[off + 64];
            __builtin_amdgcn_sched_barrier(0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[1], acc[1], 0, 0, 0);
            b0[1] = lds[off + 128]; b0[2] = lds[off + 192];
            __builtin_amdgcn_sched_barrier(0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[2], acc[2], 0, 0, 0);
            b0[3] = lds[off + 256];
            __builtin_amdgcn_sched_barrier(0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[3], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {                    // MODE 2: same, but reads interleaved between the MFMAs by the scheduler hints
        int off = tid & 63;
        float an = lds[off], bn[4];
        for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
        for (int k = 0; k < iters; ++k) {
            a = an; for (int t = 0; t < 4; ++t) b[t] = bn[t];
            off = (off + stride) & 4095;
            an = lds[off];
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[t] = lds[off + 64 * (t + 1)];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks_per_cu, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 65);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 65);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * 4 /*waves*/ * iters * 4 * 4096.0;
    printf("%-34s %d block/CU: %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms * 1e-9);
}

template <int MODE>
void sustained(const char* name, float* out) {
    // back-to-back launches for ~0.5 s: does the rate hold once the chip has been busy for a while?
    const int iters = 20000, grid = 512;
    hipEvent_t e[41]; for (auto& x : e) CK(hipEventCreate(&x));
    CK(hipEventRecord(e[0], 0));
    for (int l = 0; l < 40; ++l) {
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 65);
        CK(hipEventRecord(e[l + 1], 0));
    }
    CK(hipEventSynchronize(e[40]));
    printf("%s sustained, TFLOP/s per 3 launches:", name);
    for (int l = 0; l < 40; l += 3) {
        float ms; CK(hipEventElapsedTime(&ms, e[l], e[l + 1]));
        printf(" %.0f", 3.0 * grid * 4 * iters * 4 * 4096.0 / ms * 1e-9);
    }
    printf("\n");
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    sustained<0>("bare MFMA", out);
    sustained<7>("ping-pong LDS operands", out);
    for (int b = 1; b <= 2; ++b) {
        run<0>("bare 4x MFMA", b, out);
        run<1>("LDS operands one step ahead", b, out);
        run<2>("same + sched_group_barrier", b, out);
        run<3>("loads pinned before MFMAs", b, out);
        run<4>("pinned + scalar tap walk", b, out);
        run<5>("register copies only", b, out);
        run<6>("ping-pong sets, loads before MFMAs", b, out);
        run<7>("ping-pong, loads between MFMAs", b, out);
    }
    return 0;
}
