// Recurrent part of the projected LSTM bottleneck (nn.LSTM(384, 640, 3, proj_size=384), ResUNet.py:139),
// as ONE persistent kernel per layer instead of three launches per time step.
//
// With a projection the recurrence is  gates_t = Gx_t + W_hh (W_hr hfull_{t-1})  where hfull = o * tanh(c) is
// the 640-wide pre-projection state.  The host folds the two matrices, Wrec = W_hh W_hr [4H x H], so a step
// needs a single product and a single exchange, and does everything that has no time dependency as large
// library GEMMs outside this kernel (Gx = W_ih x + b for all t before, h = W_hr hfull for all t after).
//
// Workgroup j owns hidden units 4j..4j+3: its 16 rows of Wrec (4 gates x 4 units) stay in LDS in MFMA
// operand order for all T steps, its cell state stays in registers.  Per step each of its 4 waves multiplies
// a quarter of the K range with v_mfma_f32_16x16x4_f32 (B operand = hfull_{t-1}, read straight from L2 into
// registers), the partial sums meet in LDS, 4 x batch threads apply the gate nonlinearities and write the
// new hfull rows, and the H/4 workgroups meet at a device-wide barrier (per-workgroup flag words, coherent
// loads and stores).  The barrier spin is bounded: after the limit a workgroup raises an abort flag that
// releases every other workgroup, every state row from the aborted step on is NaN and the flag stays raised in
// sync[nwg] for the caller, so the grid always drains and a failure cannot pass for a result.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The kernel is instantiated per PADDED hidden size HP (a multiple of 16: the K range is split over 4 waves in
// 16x16x4 steps): 640 for ResUNet's bottleneck, 304 for the hidden size 300 of ResUNetOnFreq / LSTMBlock
// (ResUNet.py:335, UNet_m.py:587-610).  A state slot has HP rows; rows H..HP-1 stay zero and meet zero weights.
constexpr unsigned kSpinLimit = 1u << 21;

struct LstmArgs {
    const float* wrec;     // [4H][H] folded recurrent matrix, row g*H + u
    const float* gx;       // [T][4H][BP] input contribution incl. both biases
    float* hfull;          // [T+1][HP][BP]; slot 0 = initial state, step t reads slot t and writes slot t+1
    unsigned* sync;        // [nwg] step each workgroup has finished, then [1] abort flag
    int T, BP, nwg, H;     // H: the real hidden size (gx and wrec rows are g*H + u)
};

__device__ __forceinline__ float sigmoidf_(float x) { return fast_sigmoid(x); }

// Device-wide barrier over the nwg co-resident workgroups, without a shared counter (same-address
// atomics from 160 workgroups on 8 dies serialise at the memory side): every workgroup publishes the
// step it has finished in its own flag word, and wave 0 of every workgroup polls all the flags with
// coherent loads.  The state rows themselves are written with device-coherent stores, so no cache
// write-back / invalidate is needed, only that the stores have completed before the flag is raised.
// Returns false when aborted (bounded spin).
__device__ __forceinline__ bool grid_barrier(unsigned* flags, int nwg, unsigned step, int* s_flag) {
    // Every thread waits until its own state-row stores have been acknowledged by the memory side (they are
    // write-through, device-coherent stores) BEFORE the workgroup barrier, so that the flag store that follows
    // the barrier cannot overtake them on another memory channel.  A workgroup-scope fence alone does not wait
    // for outstanding vector stores on this target.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < 64) {
        if (tid == 0) __hip_atomic_store(&flags[blockIdx.x], step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        int ok = 1;
        for (;;) {
            bool here = true;
            for (int i = tid; i < nwg; i += 64)
                here = here && __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= step;
            if (__all(here)) break;
            const unsigned stop = __hip_atomic_load(&flags[nwg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stop != 0u || ++spins > kSpinLimit) {
                if (tid == 0) __hip_atomic_store(&flags[nwg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        if (tid == 0) *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}

template <int NCT, int kLstmH>      // 16-column batch tiles: BP = 16 * NCT; padded hidden size
__global__ __launch_bounds__(256, 1) void lstm_recurrence_kernel(const LstmArgs a) {
    constexpr int BP = 16 * NCT;
    constexpr int kLstmKS = kLstmH / 16;              // 16x16x4 k-steps per wave (K split over 4 waves)
    __shared__ float s_a[kLstmH / 4 * 64];            // [k-step][lane] A operands: 16 rows x HP
    __shared__ float s_p[4][16][BP + 1];              // per-wave partial gate sums
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = blockIdx.x * 4;

    // A operand of k-step ks, lane l: Wrec[row(l&15)][4*ks + (l>>4)], row r = gate (r>>2), unit u0 + (r&3)
    for (int i = tid; i < kLstmH / 4 * 64; i += 256) {
        const int l = i & 63, ks = i >> 6, r = l & 15, k = 4 * ks + (l >> 4);
        s_a[i] = k < a.H ? a.wrec[(size_t)((r >> 2) * a.H + u0 + (r & 3)) * a.H + k] : 0.f;
    }
    // cell-update thread: unit u0 + cu, batch column cb
    const int cu = tid / BP, cb = tid - cu * BP;
    const bool cell_thread = tid < 4 * BP;
    float c = 0.f;
    __syncthreads();

    // this wave's weight operands do not change from step to step: they stay in registers for the whole sequence
    // (re-read from LDS inside the step, each read was waited for in front of the MFMAs it feeds)
    float av[kLstmKS];
#pragma unroll
    for (int ks = 0; ks < kLstmKS; ++ks) av[ks] = s_a[(wave * kLstmKS + ks) * 64 + lane];

    int t = 0;
    for (; t < a.T; ++t) {
        const float* hprev = a.hfull + (size_t)t * kLstmH * BP;
        // ---- this wave's quarter of the K range ----
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bv[kLstmKS][NCT];
#pragma unroll
        for (int ks = 0; ks < kLstmKS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[ks][ct] = __hip_atomic_load(&hprev[(size_t)(4 * (wave * kLstmKS + ks) + (lane >> 4)) * BP + ct * 16 + (lane & 15)],
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // this step's input contribution, fetched while the MFMAs run
        float gxv[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell_thread) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gxv[g] = a.gx[((size_t)t * 4 * a.H + g * a.H + u0 + cu) * BP + cb];
        }
#pragma unroll
        for (int ks = 0; ks < kLstmKS; ++ks) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks][ct], acc[ct], 0, 0, 0);
        }
        // D: col = lane&15, row = 4*(lane>>4) + i
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_p[wave][4 * (lane >> 4) + i][ct * 16 + (lane & 15)] = acc[ct][i];
        __syncthreads();
        // ---- gates, cell, new state ----
        if (cell_thread) {
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = gxv[g] + ((s_p[0][g * 4 + cu][cb] + s_p[1][g * 4 + cu][cb]) + (s_p[2][g * 4 + cu][cb] + s_p[3][g * 4 + cu][cb]));
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = fast_tanh(pre[2]), og = sigmoidf_(pre[3]);
            c = fg * c + ig * gg;
            __hip_atomic_store(&a.hfull[((size_t)(t + 1) * kLstmH + u0 + cu) * BP + cb], og * fast_tanh(c), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!grid_barrier(a.sync, a.nwg, (unsigned)(t + 1), &s_flag)) break;
    }
    // Aborted at step t: every state row from that step on is NaN, so nothing computed from a state that was
    // never exchanged can pass for a result (sync[nwg] stays raised for the caller as well).
    if (t < a.T && cell_thread)
        for (int tt = t + 1; tt <= a.T; ++tt)
            a.hfull[((size_t)tt * kLstmH + u0 + cu) * BP + cb] = __builtin_nanf("");
}

template <int HP>
static int launch_lstm(const LstmArgs& a, int cus, hipStream_t s) {
    // the device-wide barrier needs every workgroup resident at once: ask the runtime how many workgroups of this
    // kernel (its registers and LDS) one compute unit holds, instead of assuming one
    int per_cu = 0;
    hipError_t oe;
    switch (a.BP / 16) {
        case 1: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_recurrence_kernel<1, HP>, 256, 0); break;
        case 2: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_recurrence_kernel<2, HP>, 256, 0); break;
        case 3: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_recurrence_kernel<3, HP>, 256, 0); break;
        default: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_recurrence_kernel<4, HP>, 256, 0); break;
    }
    if (oe != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_recurrence occupancy query: ") + hipGetErrorString(oe));
    if ((long long)per_cu * cus < a.nwg)
        return api_fail(UNETMI_E_UNSUPPORTED, "lstm_recurrence: the device cannot hold all workgroups at once");
    hipError_t e = hipMemsetAsync(a.sync, 0, (size_t)(a.nwg + 1) * sizeof(unsigned), s);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_recurrence memset: ") + hipGetErrorString(e));
    switch (a.BP / 16) {
        case 1: hipLaunchKernelGGL((lstm_recurrence_kernel<1, HP>), dim3(a.nwg), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((lstm_recurrence_kernel<2, HP>), dim3(a.nwg), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((lstm_recurrence_kernel<3, HP>), dim3(a.nwg), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((lstm_recurrence_kernel<4, HP>), dim3(a.nwg), dim3(256), 0, s, a); break;
    }
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("lstm_recurrence launch: ") + hipGetErrorString(e));
}

}  // namespace unetmi

using namespace unetmi;

extern "C" int unetmi_lstm_recurrence(int T, int BP, int H, const float* wrec, const float* gx, float* hfull,
                                      void* sync_ws, void* stream) {
    if (T < 0 || BP < 1 || H < 1) return api_fail(UNETMI_E_BAD_DESC, "lstm_recurrence: bad sizes");
    if (H != 640 && H != 300) return api_fail(UNETMI_E_UNSUPPORTED, "lstm_recurrence: built for hidden sizes 640 and 300");
    if (BP % 16 != 0 || BP > 64) return api_fail(UNETMI_E_UNSUPPORTED, "lstm_recurrence: padded batch must be 16, 32, 48 or 64");
    if (T == 0) return UNETMI_OK;
    if (!wrec || !gx || !hfull || !sync_ws) return api_fail(UNETMI_E_BAD_ARG, "lstm_recurrence: null pointer");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return api_fail(UNETMI_E_HIP, "lstm_recurrence: cannot query the device");
    LstmArgs a{};
    a.wrec = wrec; a.gx = gx; a.hfull = hfull; a.sync = static_cast<unsigned*>(sync_ws);
    a.T = T; a.BP = BP; a.nwg = H / 4; a.H = H;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return H == 640 ? launch_lstm<640>(a, cus, s) : launch_lstm<304>(a, cus, s);
}
