// Recurrent part of the GRU bottlenecks (DSUNet.py:281-312 TGRUBlock: nn.GRU(128, 128); UDSS.py TGRUBlock /
// cRNN: nn.GRU(256, 512), nn.GRU(256, 256)) as ONE persistent kernel per layer instead of several launches per
// time step, in the mould of the projected-LSTM kernel (lstm.hip):
//
//     r = sigmoid(gx_r + W_hr h + b_hr)     z = sigmoid(gx_z + W_hz h + b_hz)
//     n = tanh(gx_n + r * (W_hn h + b_hn))  h' = (1 - z) * n + z * h
//
// gx = W_ih x + b_ih for all steps is one large product outside this kernel.  Workgroup j owns hidden units
// 4j..4j+3: its 12 rows of W_hh (3 gates x 4 units, padded to the 16 rows of v_mfma_f32_16x16x4_f32) stay in LDS
// in operand order for all T steps; per step each of its 4 waves multiplies a quarter of the K range (B operand
// = h_{t-1}, read from L2 with device-coherent loads), the partial sums meet in LDS, 4 x batch threads apply the
// gates and write the new state rows, and the H/4 workgroups meet at a device-wide barrier of per-workgroup flag
// words.  The barrier spin is bounded: after the limit the workgroup raises an abort flag that releases all
// others, every state row from the aborted step on is NaN and the flag stays raised in sync[nwg] for the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

typedef float gru_f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kGruSpinLimit = 1u << 21;

struct GruArgs {
    const float* whh;      // [3H][H], gate order r, z, n (torch's weight_hh_l0)
    const float* bhh;      // [3H]
    const float* gx;       // input contribution incl. b_ih: element (t, row, b) at t*gx_step + row*gx_row + b
    long long gx_step, gx_row;   // [T][3H][BP]: 3H*BP, BP;  channel-major [3H][T][BP]: BP, T*BP
    float* hfull;          // [T+1][H][BP]; slot 0 = initial state, step t reads slot t and writes slot t+1
    unsigned* sync;        // [nwg] step each workgroup has finished, then [1] abort flag
    int T, BP, nwg;
    float* gates;          // training form: [T][4H][BP] rows r, z, q = W_hn h + b_hn, n of every step
};

__device__ __forceinline__ float gru_sigmoid(float x) { return fast_sigmoid(x); }

// device-wide barrier over co-resident workgroups: own flag word per workgroup, wave 0 polls them all
__device__ __forceinline__ bool gru_grid_barrier(unsigned* flags, int nwg, unsigned step, int* s_flag) {
    // the state-row stores must have been acknowledged before the flag is raised
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
            if (stop != 0u || ++spins > kGruSpinLimit) {
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

// SAVE: also keep what backpropagation through time needs (r, z, q, n of every step)
template <int NCT, int H, bool SAVE>      // 16-column batch tiles: BP = 16 * NCT; hidden size (a multiple of 16)
__global__ __launch_bounds__(256, 1) void gru_recurrence_kernel(const GruArgs a) {
    constexpr int BP = 16 * NCT;
    constexpr int KS = H / 16;                        // 16x16x4 k-steps per wave (K split over 4 waves)
    __shared__ float s_a[H / 4 * 64];                 // [k-step][lane] A operands: 16 rows (12 used) x H
    __shared__ float s_p[4][16][BP + 1];              // per-wave partial sums of W_hh h
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = blockIdx.x * 4;

    // A operand of k-step ks, lane l: W_hh[row(l&15)][4*ks + (l>>4)], row r = gate (r>>2) of unit u0 + (r&3); rows 12..15 zero
    for (int i = tid; i < H / 4 * 64; i += 256) {
        const int l = i & 63, ks = i >> 6, r = l & 15;
        s_a[i] = r < 12 ? a.whh[(size_t)((r >> 2) * H + u0 + (r & 3)) * H + 4 * ks + (l >> 4)] : 0.f;
    }
    // gate thread: unit u0 + cu, batch column cb; it carries that element of the state in a register
    const int cu = tid / BP, cb = tid - cu * BP;
    const bool gate_thread = tid < 4 * BP;
    float bh[3] = {0.f, 0.f, 0.f};
    float h = 0.f;
    if (gate_thread) {
#pragma unroll
        for (int g = 0; g < 3; ++g) bh[g] = a.bhh[g * H + u0 + cu];
        h = a.hfull[(size_t)(u0 + cu) * BP + cb];
    }
    __syncthreads();

    int t = 0;
    for (; t < a.T; ++t) {
        const float* hprev = a.hfull + (size_t)t * H * BP;
        gru_f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = gru_f32x4{0.f, 0.f, 0.f, 0.f};
        float bv[KS][NCT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                bv[ks][ct] = __hip_atomic_load(&hprev[(size_t)(4 * (wave * KS + ks) + (lane >> 4)) * BP + ct * 16 + (lane & 15)],
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float gxv[3] = {0.f, 0.f, 0.f};               // this step's input contribution, fetched while the MFMAs run
        if (gate_thread) {
#pragma unroll
            for (int g = 0; g < 3; ++g) gxv[g] = a.gx[(size_t)t * a.gx_step + (size_t)(g * H + u0 + cu) * a.gx_row + cb];
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float av = s_a[(wave * KS + ks) * 64 + lane];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ks][ct], acc[ct], 0, 0, 0);
        }
        // D: col = lane&15, row = 4*(lane>>4) + i
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_p[wave][4 * (lane >> 4) + i][ct * 16 + (lane & 15)] = acc[ct][i];
        __syncthreads();
        if (gate_thread) {
            float gh[3];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gh[g] = bh[g] + ((s_p[0][g * 4 + cu][cb] + s_p[1][g * 4 + cu][cb]) + (s_p[2][g * 4 + cu][cb] + s_p[3][g * 4 + cu][cb]));
            const float r = gru_sigmoid(gxv[0] + gh[0]), z = gru_sigmoid(gxv[1] + gh[1]);
            const float n = fast_tanh(gxv[2] + r * gh[2]);
            h = (1.f - z) * n + z * h;
            __hip_atomic_store(&a.hfull[((size_t)(t + 1) * H + u0 + cu) * BP + cb], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (SAVE) {
                float* gt = a.gates + ((size_t)t * 4 * H + u0 + cu) * BP + cb;
                gt[0] = r; gt[(size_t)H * BP] = z; gt[(size_t)2 * H * BP] = gh[2]; gt[(size_t)3 * H * BP] = n;
            }
        }
        if (!gru_grid_barrier(a.sync, a.nwg, (unsigned)(t + 1), &s_flag)) break;
    }
    if (t < a.T && gate_thread)
        for (int tt = t + 1; tt <= a.T; ++tt) a.hfull[((size_t)tt * H + u0 + cu) * BP + cb] = __builtin_nanf("");
}

template <int H, bool SAVE>
static int launch_gru(const GruArgs& a, int cus, hipStream_t s) {
    int per_cu = 0;
    hipError_t oe;
    switch (a.BP / 16) {
        case 1: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_recurrence_kernel<1, H, SAVE>, 256, 0); break;
        case 2: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_recurrence_kernel<2, H, SAVE>, 256, 0); break;
        case 3: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_recurrence_kernel<3, H, SAVE>, 256, 0); break;
        default: oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_recurrence_kernel<4, H, SAVE>, 256, 0); break;
    }
    if (oe != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("gru_recurrence occupancy query: ") + hipGetErrorString(oe));
    if ((long long)per_cu * cus < a.nwg)
        return api_fail(UNETMI_E_UNSUPPORTED, "gru_recurrence: the device cannot hold all workgroups at once");
    hipError_t e = hipMemsetAsync(a.sync, 0, (size_t)(a.nwg + 1) * sizeof(unsigned), s);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("gru_recurrence memset: ") + hipGetErrorString(e));
    switch (a.BP / 16) {
        case 1: hipLaunchKernelGGL((gru_recurrence_kernel<1, H, SAVE>), dim3(a.nwg), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((gru_recurrence_kernel<2, H, SAVE>), dim3(a.nwg), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((gru_recurrence_kernel<3, H, SAVE>), dim3(a.nwg), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL((gru_recurrence_kernel<4, H, SAVE>), dim3(a.nwg), dim3(256), 0, s, a); break;
    }
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("gru_recurrence launch: ") + hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------------------------
// One step of a GRU over MANY short sequences (the bidirectional frequency-axis GRU of UDSS.py:15-37 / UNet_m.py:
// 615-638: B*T sequences of two or a few steps): the recurrent product of a step is then one large matrix product
// (a library GEMM by the caller), and what is left is this elementwise kernel,
//     r = sigmoid(gx_r + gh_r + b_hr)   z = sigmoid(gx_z + gh_z + b_hz)   n = tanh(gx_n + r (gh_n + b_hn))
//     h' = (1 - z) n + z h,
// with row strides so that gx is read out of the all-steps / both-directions product and h' lands in the caller's
// [M, L, 2H] output directly.  gh == null and hprev == null: the first step from a zero state.
__global__ __launch_bounds__(256) void gru_gate_kernel(long long M, int H, const float* __restrict__ gx, long long gx_stride,
                                                       const float* __restrict__ gh, const float* __restrict__ bhh,
                                                       const float* __restrict__ hprev, long long hprev_stride,
                                                       float* __restrict__ out, long long out_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= M * H) return;
    const long long m = i / H;
    const int u = (int)(i - m * H);
    const float* gxr = gx + m * gx_stride;
    float hr = bhh[u], hz = bhh[H + u], hn = bhh[2 * H + u];
    if (gh) {
        const float* g = gh + m * 3 * (long long)H;
        hr += g[u]; hz += g[H + u]; hn += g[2 * H + u];
    }
    const float r = gru_sigmoid(gxr[u] + hr), z = gru_sigmoid(gxr[H + u] + hz);
    const float n = fast_tanh(gxr[2 * H + u] + r * hn);
    const float h = hprev ? hprev[m * hprev_stride + u] : 0.f;
    out[m * out_stride + u] = (1.f - z) * n + z * h;
}

// ---------------------------------------------------------------------------------------------------------------
// Backpropagation through time.  With dh_t the gradient reaching h_t (from the output and from step t+1),
//     dn = dh (1 - z),  dz = dh (h_{t-1} - n),  a_n = dn (1 - n^2),  a_r = a_n q r (1 - r),  a_z = dz z (1 - z),  dq = a_n r,
//     dh_{t-1} = dh z + W_hh^T [a_r; a_z; dq].
// Workgroup j owns hidden units 16j..16j+15: its 16 rows of W_hh^T (16 x 3H) stay in registers as MFMA A operands,
// the K range (3H) is split over the 4 waves, the B operand is [a_r; a_z; dq] of step t+1, which every workgroup
// reads back from the gates buffer (device-coherent loads) after a device-wide barrier.  The four gradients replace
// r, z, q, n of step t in place: rows [0, 3H) are the gradient of the recurrent-side pre-activations (W_hh h + b_hh),
// rows [0, 2H) and [3H, 4H) that of the input-side ones, for the time-batched weight-gradient products afterwards.
struct GruBwdArgs {
    const float* whh;      // [3H][H]
    float* gates;          // [T][4H][BP]
    const float* hfull;    // [T+1][H][BP]
    const float* dy;       // [T][H][BP] gradient of the loss w.r.t. the outputs h_1..h_T
    float* dh0;            // [H][BP] gradient reaching the initial state
    unsigned* sync;
    int T, BP, nwg;
};

template <int NCT, int H>
__global__ __launch_bounds__(256, 1) void gru_bptt_kernel(const GruBwdArgs a) {
    constexpr int BP = 16 * NCT;
    constexpr int KS = 3 * H / 16;                    // 16x16x4 k-steps per wave
    constexpr int KC = 24;                            // k-steps whose B operands are in flight together
    static_assert(KS % KC == 0, "k-steps come in whole chunks");
    __shared__ float s_p[4][16][BP + 1];
    __shared__ int s_flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = blockIdx.x * 16;

    // A operand of k-step ks, lane l: W_hh^T[u0 + (l&15)][k] = W_hh[k][u0 + (l&15)], k = 4*(wave*KS + ks) + (l>>4)
    float av[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) av[ks] = a.whh[(size_t)(4 * (wave * KS + ks) + (lane >> 4)) * H + u0 + (lane & 15)];

    // gate elements of this thread: e = tid + 256*j -> unit u0 + e / BP, column e % BP
    float carry[NCT];                                 // dh_{t+1} * z_{t+1}
#pragma unroll
    for (int j = 0; j < NCT; ++j) carry[j] = 0.f;

    int s = 0;
    for (; s <= a.T; ++s) {
        const int t = a.T - 1 - s;                    // t = -1: only the gradient of the initial state is left
        float r[NCT], z[NCT], q[NCT], n[NCT], hp[NCT], dyv[NCT];
        if (t >= 0) {
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int e = tid + 256 * j, cu = e / BP, cb = e - cu * BP;
                const float* gt = a.gates + ((size_t)t * 4 * H + u0 + cu) * BP + cb;
                r[j] = gt[0]; z[j] = gt[(size_t)H * BP]; q[j] = gt[(size_t)2 * H * BP]; n[j] = gt[(size_t)3 * H * BP];
                hp[j] = a.hfull[((size_t)t * H + u0 + cu) * BP + cb];
                dyv[j] = a.dy[((size_t)t * H + u0 + cu) * BP + cb];
            }
        }
        gru_f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = gru_f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {                                  // W_hh^T [a_r; a_z; dq]_{t+1}: this wave's quarter of the K range
            const float* dprev = a.gates + (size_t)(t + 1) * 4 * H * BP;
#pragma unroll
            for (int kc = 0; kc < KS / KC; ++kc) {
                float bv[KC][NCT];
#pragma unroll
                for (int ks = 0; ks < KC; ++ks)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        bv[ks][ct] = __hip_atomic_load(&dprev[(size_t)(4 * (wave * KS + kc * KC + ks) + (lane >> 4)) * BP + ct * 16 + (lane & 15)],
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int ks = 0; ks < KC; ++ks)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc * KC + ks], bv[ks][ct], acc[ct], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 4; ++i) s_p[wave][4 * (lane >> 4) + i][ct * 16 + (lane & 15)] = acc[ct][i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int e = tid + 256 * j, cu = e / BP, cb = e - cu * BP;
            const float back = (s_p[0][cu][cb] + s_p[1][cu][cb]) + (s_p[2][cu][cb] + s_p[3][cu][cb]);
            if (t < 0) {
                a.dh0[(size_t)(u0 + cu) * BP + cb] = carry[j] + back;
                continue;
            }
            const float dh = dyv[j] + carry[j] + back;
            const float an = dh * (1.f - z[j]) * (1.f - n[j] * n[j]);
            const float ar = an * q[j] * r[j] * (1.f - r[j]);
            const float az = dh * (hp[j] - n[j]) * z[j] * (1.f - z[j]);
            carry[j] = dh * z[j];
            float* gt = a.gates + ((size_t)t * 4 * H + u0 + cu) * BP + cb;
            __hip_atomic_store(&gt[0], ar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gt[(size_t)H * BP], az, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&gt[(size_t)2 * H * BP], an * r[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gt[(size_t)3 * H * BP] = an;
        }
        if (t < 0) break;
        if (!gru_grid_barrier(a.sync, a.nwg, (unsigned)(s + 1), &s_flag)) break;
    }
    // Aborted: every gradient from that step down is NaN (sync[nwg] stays raised for the caller as well)
    if (s < a.T) {
        for (int ss = s; ss < a.T; ++ss)
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int e = tid + 256 * j, cu = e / BP, cb = e - cu * BP;
                float* gt = a.gates + ((size_t)(a.T - 1 - ss) * 4 * H + u0 + cu) * BP + cb;
#pragma unroll
                for (int g = 0; g < 4; ++g) gt[(size_t)g * H * BP] = __builtin_nanf("");
            }
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int e = tid + 256 * j, cu = e / BP, cb = e - cu * BP;
            a.dh0[(size_t)(u0 + cu) * BP + cb] = __builtin_nanf("");
        }
    }
}

template <int NCT, int H>
static int launch_gru_bptt_n(const GruBwdArgs& a, int cus, hipStream_t s) {
    int per_cu = 0;
    const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gru_bptt_kernel<NCT, H>, 256, 0);
    if (oe != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("gru_bptt occupancy query: ") + hipGetErrorString(oe));
    if ((long long)per_cu * cus < a.nwg) return api_fail(UNETMI_E_UNSUPPORTED, "gru_bptt: the device cannot hold all workgroups at once");
    hipError_t e = hipMemsetAsync(a.sync, 0, (size_t)(a.nwg + 1) * sizeof(unsigned), s);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("gru_bptt memset: ") + hipGetErrorString(e));
    hipLaunchKernelGGL((gru_bptt_kernel<NCT, H>), dim3(a.nwg), dim3(256), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("gru_bptt launch: ") + hipGetErrorString(e));
}

template <int H>
static int launch_gru_bptt(const GruBwdArgs& a, int cus, hipStream_t s) {
    switch (a.BP / 16) {
        case 1: return launch_gru_bptt_n<1, H>(a, cus, s);
        case 2: return launch_gru_bptt_n<2, H>(a, cus, s);
        case 3: return launch_gru_bptt_n<3, H>(a, cus, s);
        default: return launch_gru_bptt_n<4, H>(a, cus, s);
    }
}

}  // namespace unetmi

using namespace unetmi;

static int gru_run(int T, int BP, int H, const float* whh, const float* bhh, const float* gx, int gx_channel_major,
                   float* hfull, float* gates, void* sync_ws, void* stream);

extern "C" int unetmi_gru_gate_fwd(long long M, int H, const float* gx, long long gx_stride, const float* gh, const float* bhh,
                                   const float* hprev, long long hprev_stride, float* out, long long out_stride, void* stream) {
    if (M < 0 || H < 1 || gx_stride < 3ll * H || out_stride < H || (hprev && hprev_stride < H))
        return api_fail(UNETMI_E_BAD_DESC, "gru_gate: bad sizes");
    if (M == 0) return UNETMI_OK;
    if (!gx || !bhh || !out) return api_fail(UNETMI_E_BAD_ARG, "gru_gate: null pointer");
    const long long blocks = (M * H + 255) / 256;
    if (blocks > 0x7fffffffll) return api_fail(UNETMI_E_UNSUPPORTED, "gru_gate: too many elements");
    hipLaunchKernelGGL(gru_gate_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), M, H, gx, gx_stride,
                       gh, bhh, hprev, hprev_stride, out, out_stride);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("gru_gate launch: ") + hipGetErrorString(e));
}

extern "C" int unetmi_gru_recurrence(int T, int BP, int H, const float* whh, const float* bhh, const float* gx,
                                     int gx_channel_major, float* hfull, void* sync_ws, void* stream) {
    return gru_run(T, BP, H, whh, bhh, gx, gx_channel_major, hfull, nullptr, sync_ws, stream);
}

extern "C" int unetmi_gru_recurrence_train(int T, int BP, int H, const float* whh, const float* bhh, const float* gx,
                                           float* hfull, float* gates, void* sync_ws, void* stream) {
    if (!gates) return api_fail(UNETMI_E_BAD_ARG, "gru_recurrence_train: null pointer");
    return gru_run(T, BP, H, whh, bhh, gx, 0, hfull, gates, sync_ws, stream);
}

extern "C" int unetmi_gru_bptt(int T, int BP, int H, const float* whh, float* gates, const float* hfull, const float* dy,
                               float* dh0, void* sync_ws, void* stream) {
    if (T < 0 || BP < 1 || H < 1) return api_fail(UNETMI_E_BAD_DESC, "gru_bptt: bad sizes");
    if (H != 128 && H != 256 && H != 512) return api_fail(UNETMI_E_UNSUPPORTED, "gru_bptt: built for hidden sizes 128, 256 and 512");
    if (BP % 16 != 0 || BP > 64) return api_fail(UNETMI_E_UNSUPPORTED, "gru_bptt: padded batch must be 16, 32, 48 or 64");
    if (!whh || !dh0 || !sync_ws || (T > 0 && (!gates || !hfull || !dy))) return api_fail(UNETMI_E_BAD_ARG, "gru_bptt: null pointer");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return api_fail(UNETMI_E_HIP, "gru_bptt: cannot query the device");
    GruBwdArgs a{};
    a.whh = whh; a.gates = gates; a.hfull = hfull; a.dy = dy; a.dh0 = dh0; a.sync = static_cast<unsigned*>(sync_ws);
    a.T = T; a.BP = BP; a.nwg = H / 16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (H == 128) return launch_gru_bptt<128>(a, cus, s);
    if (H == 256) return launch_gru_bptt<256>(a, cus, s);
    return launch_gru_bptt<512>(a, cus, s);
}

static int gru_run(int T, int BP, int H, const float* whh, const float* bhh, const float* gx, int gx_channel_major,
                   float* hfull, float* gates, void* sync_ws, void* stream) {
    if (T < 0 || BP < 1 || H < 1) return api_fail(UNETMI_E_BAD_DESC, "gru_recurrence: bad sizes");
    if (H != 128 && H != 256 && H != 512) return api_fail(UNETMI_E_UNSUPPORTED, "gru_recurrence: built for hidden sizes 128, 256 and 512");
    if (BP % 16 != 0 || BP > 64) return api_fail(UNETMI_E_UNSUPPORTED, "gru_recurrence: padded batch must be 16, 32, 48 or 64");
    if (T == 0) return UNETMI_OK;
    if (!whh || !bhh || !gx || !hfull || !sync_ws) return api_fail(UNETMI_E_BAD_ARG, "gru_recurrence: null pointer");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return api_fail(UNETMI_E_HIP, "gru_recurrence: cannot query the device");
    GruArgs a{};
    a.whh = whh; a.bhh = bhh; a.gx = gx; a.hfull = hfull; a.sync = static_cast<unsigned*>(sync_ws);
    a.T = T; a.BP = BP; a.nwg = H / 4; a.gates = gates;
    a.gx_step = gx_channel_major ? BP : (long long)3 * H * BP;
    a.gx_row = gx_channel_major ? (long long)T * BP : BP;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (gates) {
        if (H == 128) return launch_gru<128, true>(a, cus, s);
        if (H == 256) return launch_gru<256, true>(a, cus, s);
        return launch_gru<512, true>(a, cus, s);
    }
    if (H == 128) return launch_gru<128, false>(a, cus, s);
    if (H == 256) return launch_gru<256, false>(a, cus, s);
    return launch_gru<512, false>(a, cus, s);
}
