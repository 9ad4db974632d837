// Recurrent part of the projected LSTM bottleneck (nn.LSTM(384, 640, 3, proj_size=384), ResUNet.py:139) with the
// state exchange kept INSIDE each XCD.
//
// lstm.hip gives every workgroup four hidden units for all batch columns, so each step every workgroup reads the
// whole state (80 KB) with device-coherent loads: 13 MB per step cross the fabric and a step takes 6-7 us.  Here
// the BATCH is split over the chip's 8 XCDs instead: XCD x owns batch columns [4x, 4x+4) and runs a complete,
// independent recurrence for them -- no XCD ever reads another's state, so the exchange is served by the XCD's
// own L2 (device-scope loads, which skip the per-CU cache) and there is no device-wide barrier at all.
//
// Inside an XCD the 32 workgroups (one per compute unit) split the hidden units, 20 each = 80 rows of the folded
// recurrent matrix Wrec = W_hh W_hr [4H x H].  A workgroup keeps its 80 x 640 slice in REGISTERS for all T steps:
// wave w holds columns [160w, 160w+160) as 200 A operands of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4x1
// blocks per instruction: 4 hidden units x 4 values of k; a block's 4 rows are the 4 gates of one unit, its 4
// columns the XCD's 4 batch columns).  Per step a wave
//   * reads its quarter of the state h_t (640 values) as 8-byte {value, step} granules, written by one store
//     and read by one load each, so a granule is either old or new, never torn, and carries its own readiness:
//     no flag, no wait for store completion on the producer side; the wave retries until all its tags read t;
//   * issues 200 MFMAs, sums the 4 k-blocks of a unit with two DPP adds and leaves its partial sums in LDS;
// then (one workgroup barrier) 80 threads add the four waves' partials and the input contribution, apply the
// gate nonlinearities, keep the cell state in a register, and publish the new state granules (double-buffered
// by step parity) and the state row of the caller's output.
//
// Placement: a workgroup reads the XCD it actually runs on from the hardware (XCC_ID) and takes a ticket for a
// slot of that XCD, so correctness never depends on the dispatcher's round-robin; if an XCD ends up short of
// workgroups (the GPU is shared), the bounded spin raises the abort word, everything from that step on is NaN
// and the status word stays non-zero for the caller.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The kernels are instantiated per PADDED hidden size HP: 640 (ResUNet's bottleneck: 32 workgroups per XCD) and 320
// (hidden size 300 of ResUNetOnFreq / LSTMBlock, ResUNet.py:335, UNet_m.py:587-610: 16 workgroups per XCD).  Units
// H..HP-1 meet zero weights, publish zeros and touch no caller tensor.
constexpr int kXHMax = 640;              // largest padded hidden size (workspace sizing)
constexpr int kXcds = 8;                 // XCDs
constexpr int kXU = 20;                  // hidden units per workgroup: HP / (workgroups per XCD)
constexpr int kXSets = kXU / 4;          // MFMA accumulator sets: 4 units each (5)
constexpr int kXCols = 4;                // batch columns per XCD
constexpr unsigned kXSpinLimit = 1u << 20;

struct LstmXcdArgs {
    const float* wrec;          // [4H][H]
    const float* gx;            // step-major [T][4H][BP], or channel-major [4H][T+1][BP] (row t+1 = step t)
    float* hfull;               // step-major [T+1][H][BP], or channel-major [H][T+1][BP]
    unsigned* ctl;              // [8] tickets, [8] abort word (+ padding to 64 words)
    unsigned long long* hx;     // [8 XCDs][2 parities][H][4] granules {float value, unsigned step}
    int T, BP, H;               // H: the real hidden size (rows of wrec / gx are g*H + u)
    long long gx_step, gx_row, gx_first;      // element strides of gx per step / per gate row, offset of step 0
    long long h_step, h_row;                  // element strides of hfull per step slot / per hidden unit
    float* gates;               // training form: the activated gates i, f, g, o of step t, laid out (and strided) as gx;
    float* cell;                //                the cell state c_t, laid out (and strided) as hfull
};

__device__ __forceinline__ float sigmoid_x(float x) { return fast_sigmoid(x); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {     // v + (v moved by the DPP control), within rows of 16 lanes
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

// SAVE: also keep what backpropagation through time needs -- the four activated gates of every step (written over
// the step's input contribution: a.gates may be a.gx) and the cell state.
// F16: the recurrent product Wrec * m_{t-1} takes fp16 operands (v_mfma_f32_4x4x4_16b_f16: four k per instruction, a
// quarter of the 200 fp32 instructions of a step; fp32 accumulation, gates, cell and published state stay fp32).  For
// callers whose input was rounded to 16 bits anyway: |m| < 1 and the folded matrix is O(1/sqrt(H)), well inside fp16.
template <bool SAVE, int kXH, bool F16 = false>
__global__ __launch_bounds__(256, 1) void lstm_xcd_kernel(const LstmXcdArgs a) {
    static_assert(!(SAVE && F16), "the training form keeps fp32 operands");
    constexpr int kXWg = kXH / kXU;          // workgroups (compute units) used per XCD
    constexpr int kXKg = kXH / 4 / 4;        // k-groups per wave: HP/4 columns / 4 per instruction
    constexpr int kXLd = kXH / 64;           // state granules a lane reads per step
    __shared__ float s_h[4][kXH];                          // per wave: its quarter of the state, [k][col]
    __shared__ float s_part[2][4][kXSets][4][16];          // [parity][wave][set][gate][unit-in-set * 4 + col]
    __shared__ int s_slot;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- where am I: XCD from the hardware, slot from a per-XCD ticket ----
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    if (tid == 0) s_slot = (int)__hip_atomic_fetch_add(&a.ctl[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int col0 = xcd * kXCols;
    if (slot >= kXWg || col0 >= a.BP) return;              // surplus workgroup, or an XCD without batch columns

    // ---- lane roles.  lane = 16*ui + 4*ki + r: A operand row r = gate r of unit ui of the set, B/D column r ----
    const int ui = lane >> 4, ki = (lane >> 2) & 3, r = lane & 3;
    const int u_base = slot * kXU;
    constexpr int kXKq = kXKg / 4;           // fp16 form: k-blocks of 4 x 4 columns per wave
    float A[F16 ? 1 : kXSets][F16 ? 1 : kXKg];
    f16x4 A16[F16 ? kXSets : 1][F16 ? kXKq : 1];      // lane (unit ui, k-group ki, gate r): columns 16*kq + 4*ki .. +3 of its wave's quarter
#pragma unroll
    for (int s = 0; s < kXSets; ++s) {
        const int u = u_base + 4 * s + ui;
        if constexpr (F16) {
            const int k0 = kXH / 4 * wave + 4 * ki;
            const float* row = a.wrec + (size_t)(r * a.H + u) * a.H + k0;
#pragma unroll
            for (int kq = 0; kq < kXKq; ++kq)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    A16[s][kq][e] = (_Float16)((u < a.H && k0 + 16 * kq + e < a.H) ? row[16 * kq + e] : 0.f);
        } else {
            const int k0 = kXH / 4 * wave + ki;
            const float* row = a.wrec + (size_t)(r * a.H + u) * a.H + k0;
#pragma unroll
            for (int kg = 0; kg < kXKg; ++kg) A[s][kg] = (u < a.H && k0 + 4 * kg < a.H) ? row[4 * kg] : 0.f;
        }
    }
    // gate thread: cell (unit ul, column cj) of this workgroup; a padding unit (>= H) only publishes zeros
    const bool gate_thread = tid < kXU * kXCols;
    const int ul = tid >> 2, cj = tid & 3;
    const bool col_live = col0 + cj < a.BP && u_base + ul < a.H;
    float c = 0.f;

    unsigned long long* hx = a.hx + (size_t)xcd * 2 * kXH * kXCols;
    float* sh = s_h[wave];
    bool alive = true;
    int t = 0;
    for (; t < a.T; ++t) {
        // this step's input contribution, fetched while the state is awaited
        float gxv[4] = {0.f, 0.f, 0.f, 0.f};
        if (gate_thread && col_live) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                gxv[g] = a.gx[a.gx_first + t * a.gx_step + (g * a.H + u_base + ul) * a.gx_row + col0 + cj];
        }
        // ---- this wave's quarter of h_t: granules (HP/4*wave + e/4, e%4), e = lane + 64*i, until all read step t ----
        const unsigned long long* src = hx + (size_t)(t & 1) * kXH * kXCols + (size_t)wave * (kXH / 4) * kXCols;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            float v[kXLd];
#pragma unroll
            for (int i = 0; i < kXLd; ++i) {
                // device-scope loads: they never hit this compute unit's vector cache (a line cached before the
                // producers' stores would otherwise be hit again and again) and are served by the XCD's L2, where
                // those stores land.  Measured alternatives: invalidating the vector cache before plain loads is 3x
                // slower per step, non-temporal loads the same speed.
                const unsigned long long g =
                    __hip_atomic_load(&src[lane + 64 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[i] = __builtin_bit_cast(float, (unsigned)(g & 0xffffffffull));
                ok = ok && (unsigned)(g >> 32) == (unsigned)t;
            }
            if (__all(ok)) {
                if constexpr (F16) {                 // [column][k] halves: a lane's four k of one column are one 8-byte read
                    _Float16* sh16 = reinterpret_cast<_Float16*>(sh);
#pragma unroll
                    for (int i = 0; i < kXLd; ++i) {
                        const int e = lane + 64 * i;
                        sh16[(e & 3) * (kXH / 4) + (e >> 2)] = (_Float16)v[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kXLd; ++i) sh[lane + 64 * i] = v[i];
                }
                break;
            }
            if ((++spins & 63u) == 0u) {
                const unsigned stop = __hip_atomic_load(&a.ctl[kXcds], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (stop != 0u || spins > kXSpinLimit) {
                    if (lane == 0) __hip_atomic_store(&a.ctl[kXcds], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    alive = false;
                    break;
                }
            }
        }
        // ---- 200 MFMAs: acc[s] block (ui, ki) += A rows (4 gates of unit 4s+ui) x B (h[k][0..3]), k = 160w + 4kg + ki ----
        f32x4 acc[kXSets];
#pragma unroll
        for (int s = 0; s < kXSets; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F16) {
            const _Float16* sh16 = reinterpret_cast<const _Float16*>(sh) + r * (kXH / 4) + 4 * ki;
            f16x4 bq[kXKq];
#pragma unroll
            for (int kq = 0; kq < kXKq; ++kq) bq[kq] = *reinterpret_cast<const f16x4*>(sh16 + 16 * kq);
#pragma unroll
            for (int kq = 0; kq < kXKq; ++kq)
#pragma unroll
                for (int s = 0; s < kXSets; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_4x4x4f16(A16[s][kq], bq[kq], acc[s], 0, 0, 0);
        } else {
        // The state operands are read a group of 8 k-blocks ahead into a second register set: one read waited for
        // in front of each 10 MFMAs (80 cycles) put ~19 LDS round trips on the critical path of every time step.
        constexpr int kGrp = kXKg % 8 == 0 ? 8 : 4;
        static_assert(kXKg % kGrp == 0, "k-blocks come in whole groups");
        float bv[2][kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) bv[0][q] = sh[(4 * q + ki) * kXCols + r];
#pragma unroll
        for (int gi = 0; gi < kXKg / kGrp; ++gi) {
            const int cur = gi & 1;
            if (gi + 1 < kXKg / kGrp) {
#pragma unroll
                for (int q = 0; q < kGrp; ++q) bv[cur ^ 1][q] = sh[(4 * ((gi + 1) * kGrp + q) + ki) * kXCols + r];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < kGrp; ++q)
#pragma unroll
                for (int s = 0; s < kXSets; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[s][gi * kGrp + q], bv[cur][q], acc[s], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // sum the 4 k-blocks of a unit (lanes 4*ki apart inside a row of 16), leave the partials in LDS
        float* part = &s_part[t & 1][wave][0][0][0];
#pragma unroll
        for (int s = 0; s < kXSets; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x = acc[s][g];
                x = dpp_add<0x128>(x);          // row_ror:8
                x = dpp_add<0x124>(x);          // row_ror:4
                if (ki == 0) part[(s * 4 + g) * 16 + ui * 4 + r] = x;
            }
        if (!__syncthreads_and(alive ? 1 : 0)) break;       // partials visible; if some wave gave up, leave together
        // ---- gates, cell, new state ----
        if (gate_thread) {
            const int s = ul >> 2, uu = ul & 3;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int idx = (s * 4 + g) * 16 + uu * 4 + cj;
                pre[g] = gxv[g] + ((s_part[t & 1][0][0][0][idx] + s_part[t & 1][1][0][0][idx]) +
                                   (s_part[t & 1][2][0][0][idx] + s_part[t & 1][3][0][0][idx]));
            }
            const float ig = sigmoid_x(pre[0]), fg = sigmoid_x(pre[1]), gg = fast_tanh(pre[2]), og = sigmoid_x(pre[3]);
            c = fg * c + ig * gg;
            const float hn = col_live ? og * fast_tanh(c) : 0.f;
            const unsigned long long granule =
                (unsigned long long)__builtin_bit_cast(unsigned, hn) | ((unsigned long long)(unsigned)(t + 1) << 32);
            // written through the vector cache into the XCD's L2, which is where this XCD's readers look; the
            // device-scope (memory-side) form of the store costs 0.6 us more per step and is not needed: no other
            // XCD ever reads these granules
            __hip_atomic_store(&hx[(size_t)((t + 1) & 1) * kXH * kXCols + (size_t)(u_base + ul) * kXCols + cj], granule,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (col_live) a.hfull[(t + 1) * a.h_step + (u_base + ul) * a.h_row + col0 + cj] = hn;
            if (SAVE && col_live) {
                const float gv[4] = {ig, fg, gg, og};
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    a.gates[a.gx_first + t * a.gx_step + (g * a.H + u_base + ul) * a.gx_row + col0 + cj] = gv[g];
                a.cell[(t + 1) * a.h_step + (u_base + ul) * a.h_row + col0 + cj] = c;
            }
        }
    }
    // Aborted at step t: every state row from that step on is NaN so that nothing computed from a state that was
    // never exchanged can pass for a result (the abort word stays raised for the caller as well).
    if (t < a.T && gate_thread && col_live)
        for (int tt = t + 1; tt <= a.T; ++tt) a.hfull[tt * a.h_step + (u_base + ul) * a.h_row + col0 + cj] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------------------------
// Backpropagation through time for the same recurrence, same placement: XCD x owns batch columns [4x, 4x+4), its 32
// workgroups own 20 hidden units each.  With m_t = o_t * tanh(c_t) (the pre-projection state) and G_t the gradient
// of the loss with respect to step t's four gate pre-activations,
//     dm_t = dhfull_t + Wrec^T G_{t+1},      dc_t = dm_t * o_t * (1 - tanh^2 c_t) + dc_{t+1} * f_{t+1},
//     G_t  = { dc_t g_t i_t(1-i_t),  dc_t c_{t-1} f_t(1-f_t),  dc_t i_t (1-g_t^2),  dm_t tanh(c_t) o_t(1-o_t) },
// where dhfull_t = W_hr^T dL/dy_t is computed for all steps beforehand as one matrix product.  What is exchanged per
// step is G (4H = 2560 rows x 4 columns per XCD, as {value, step} granules in the XCD's L2); a workgroup's slice of
// Wrec^T (20 rows x 2560) stays in registers as 200 A operands per lane: wave w holds the rows of gate w, an
// instruction's 16 blocks are 16 consecutive values of k, a block's 4 rows are 4 of the workgroup's units, its 4
// columns the XCD's batch columns.  G_t replaces the saved gates of step t in place: the weight gradients are
// time-batched matrix products over it afterwards.
struct LstmXcdBwdArgs {
    const float* wrec;          // [4H][H]
    float* gates;               // [4H][T+1][BP]: in = activated gates of step t at slot t+1, out = G_t
    const float* cell;          // [H][T+1][BP]: c_t at slot t+1, slot 0 = 0
    const float* dhfull;        // [H][T+1][BP]: gradient reaching m_t through the projection, at slot t+1
    unsigned* ctl;              // [8] tickets, [8] abort word (+ padding to 64 words)
    unsigned long long* gxch;   // [8 XCDs][2 parities][4 HP][4] granules {float value, unsigned step}
    int T, BP, H;
};

template <int kXH>
__global__ __launch_bounds__(256, 1) void lstm_xcd_bwd_kernel(const LstmXcdBwdArgs a) {
    constexpr int kXWg = kXH / kXU;          // workgroups used per XCD
    constexpr int kXG = 4 * kXH;             // exchanged rows per step
    constexpr int kXBKg = kXH / 16;          // k-groups per wave: HP rows / 16 per instruction
    constexpr int kXBLd = kXH * kXCols / 64; // granules a lane reads per step
    __shared__ float s_g[4][kXH * kXCols];                 // per wave: its gate's rows of G, [k][col]
    __shared__ float s_part[2][4][4][kXSets][4][4];        // [parity][wave][row of 16 lanes][set][unit][col]
    __shared__ int s_slot;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int xcd = (int)(xcc & 7u);
    if (tid == 0) s_slot = (int)__hip_atomic_fetch_add(&a.ctl[xcd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int slot = s_slot;
    const int col0 = xcd * kXCols;
    if (slot >= kXWg || col0 >= a.BP) return;

    // lane = 4*b + r: block b = value of k inside a group of 16, A operand row r = unit r of the set, B/D column r
    const int b = lane >> 2, r = lane & 3;
    const int u_base = slot * kXU;
    float A[kXSets][kXBKg];
#pragma unroll
    for (int s = 0; s < kXSets; ++s) {
        const int u = u_base + 4 * s + r;
        const float* colp = a.wrec + (size_t)(a.H * wave + b) * a.H + u;
#pragma unroll
        for (int kg = 0; kg < kXBKg; ++kg) A[s][kg] = (u < a.H && b + 16 * kg < a.H) ? colp[(size_t)16 * kg * a.H] : 0.f;
    }
    const bool gate_thread = tid < kXU * kXCols;
    const int ul = tid >> 2, cj = tid & 3;
    const bool col_live = col0 + cj < a.BP && u_base + ul < a.H;
    const long long row = (long long)(a.T + 1) * a.BP;
    float dc_carry = 0.f;                                  // dc_{t+1} * f_{t+1}

    unsigned long long* gx = a.gxch + (size_t)xcd * 2 * kXG * kXCols;
    float* sg = s_g[wave];
    bool alive = true;
    int s = 0;
    for (; s < a.T; ++s) {
        const int t = a.T - 1 - s;
        // what this step needs from the forward pass, fetched while G_{t+1} is awaited
        float gv[4] = {0.f, 0.f, 0.f, 0.f}, c_t = 0.f, c_prev = 0.f, dm = 0.f;
        if (gate_thread && col_live) {
            const long long at = (long long)(t + 1) * a.BP + col0 + cj;
#pragma unroll
            for (int g = 0; g < 4; ++g) gv[g] = a.gates[(g * a.H + u_base + ul) * row + at];
            c_t = a.cell[(u_base + ul) * row + at];
            c_prev = a.cell[(u_base + ul) * row + at - a.BP];
            dm = a.dhfull[(u_base + ul) * row + at];
        }
        // ---- this wave's gate of G_{t+1}: granules (HP*wave + e/4, e%4), e = lane + 64*i, until all carry tag s ----
        const unsigned long long* src = gx + (size_t)(s & 1) * kXG * kXCols + (size_t)wave * kXH * kXCols;
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            float v[kXBLd];
#pragma unroll
            for (int i = 0; i < kXBLd; ++i) {
                const unsigned long long g =
                    __hip_atomic_load(&src[lane + 64 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[i] = __builtin_bit_cast(float, (unsigned)(g & 0xffffffffull));
                ok = ok && (unsigned)(g >> 32) == (unsigned)s;
            }
            if (__all(ok)) {
#pragma unroll
                for (int i = 0; i < kXBLd; ++i) sg[lane + 64 * i] = v[i];
                break;
            }
            if ((++spins & 63u) == 0u) {
                const unsigned stop = __hip_atomic_load(&a.ctl[kXcds], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (stop != 0u || spins > kXSpinLimit) {
                    if (lane == 0) __hip_atomic_store(&a.ctl[kXcds], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    alive = false;
                    break;
                }
            }
        }
        // ---- 200 MFMAs: acc[set] block b += A (Wrec[k][4 units of the set]) x B (G[k][0..3]), k = 640w + 16kg + b ----
        f32x4 acc[kXSets];
#pragma unroll
        for (int q = 0; q < kXSets; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int kGrp = kXBKg % 8 == 0 ? 8 : 4;
        static_assert(kXBKg % kGrp == 0, "k-groups come in whole groups");
        float bv[2][kGrp];
#pragma unroll
        for (int q = 0; q < kGrp; ++q) bv[0][q] = sg[64 * q + lane];
#pragma unroll
        for (int gi = 0; gi < kXBKg / kGrp; ++gi) {
            const int cur = gi & 1;
            if (gi + 1 < kXBKg / kGrp) {
#pragma unroll
                for (int q = 0; q < kGrp; ++q) bv[cur ^ 1][q] = sg[64 * ((gi + 1) * kGrp + q) + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < kGrp; ++q)
#pragma unroll
                for (int st = 0; st < kXSets; ++st)
                    acc[st] = __builtin_amdgcn_mfma_f32_4x4x1f32(A[st][gi * kGrp + q], bv[cur][q], acc[st], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // sum the 4 blocks inside each row of 16 lanes; the 4 rows and the 4 waves meet in LDS
        float* part = &s_part[s & 1][wave][lane >> 4][0][0][0];
#pragma unroll
        for (int st = 0; st < kXSets; ++st)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = acc[st][i];
                x = dpp_add<0x128>(x);          // row_ror:8
                x = dpp_add<0x124>(x);          // row_ror:4
                if ((lane & 12) == 0) part[(st * 4 + i) * 4 + r] = x;
            }
        if (!__syncthreads_and(alive ? 1 : 0)) break;
        // ---- this step's gate gradients ----
        if (gate_thread) {
            const int st = ul >> 2, i = ul & 3;
            float back = 0.f;                              // (Wrec^T G_{t+1})[unit][col], summed in a fixed order
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float pw = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) pw += s_part[s & 1][w][q][st][i][cj];
                back += pw;
            }
            const float dmv = dm + back;
            const float ig = gv[0], fg = gv[1], gg = gv[2], og = gv[3];
            const float tc = fast_tanh(c_t);
            const float dc = dmv * og * (1.f - tc * tc) + dc_carry;
            dc_carry = dc * fg;
            float G[4] = {dc * gg * ig * (1.f - ig), dc * c_prev * fg * (1.f - fg), dc * ig * (1.f - gg * gg),
                          dmv * tc * og * (1.f - og)};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float val = col_live ? G[g] : 0.f;
                const unsigned long long granule =
                    (unsigned long long)__builtin_bit_cast(unsigned, val) | ((unsigned long long)(unsigned)(s + 1) << 32);
                __hip_atomic_store(&gx[(size_t)((s + 1) & 1) * kXG * kXCols + (size_t)(g * kXH + u_base + ul) * kXCols + cj],
                                   granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (col_live) a.gates[(g * a.H + u_base + ul) * row + (long long)(t + 1) * a.BP + col0 + cj] = val;
            }
        }
    }
    // Aborted: every gradient from that step down is NaN (the abort word stays raised for the caller as well).
    if (s < a.T && gate_thread && col_live)
        for (int ss = s; ss < a.T; ++ss)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                a.gates[(g * a.H + u_base + ul) * row + (long long)(a.T - ss) * a.BP + col0 + cj] = __builtin_nanf("");
}

}  // namespace unetmi

using namespace unetmi;

extern "C" size_t unetmi_lstm_recurrence_xcd_workspace_bytes(void) {
    return 64 * sizeof(unsigned) + (size_t)kXcds * 2 * kXHMax * kXCols * sizeof(unsigned long long);
}

static int lstm_xcd_run(int T, int BP, int H, const float* wrec, const float* gx, float* hfull, int channel_major,
                        float* gates, float* cell, void* ws, size_t ws_bytes, void* stream);

extern "C" int unetmi_lstm_recurrence_xcd(int T, int BP, int H, const float* wrec, const float* gx, float* hfull,
                                          int channel_major, void* ws, size_t ws_bytes, void* stream) {
    return lstm_xcd_run(T, BP, H, wrec, gx, hfull, channel_major, nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int unetmi_lstm_recurrence_xcd_train(int T, int BP, int H, const float* wrec, float* gx_gates, float* hfull,
                                                float* cell, void* ws, size_t ws_bytes, void* stream) {
    if (!cell) return api_fail(UNETMI_E_BAD_ARG, "lstm_recurrence_xcd_train: null pointer");
    return lstm_xcd_run(T, BP, H, wrec, gx_gates, hfull, 1, gx_gates, cell, ws, ws_bytes, stream);
}

// what the XCD-local kernels need from the device, common to the forward and the reverse recurrence
static int lstm_xcd_device_ok(const char* who, int H, int BP) {
    if (H != 640 && H != 300) return api_fail(UNETMI_E_UNSUPPORTED, std::string(who) + ": built for hidden sizes 640 and 300");
    if (BP > kXcds * kXCols) return api_fail(UNETMI_E_UNSUPPORTED, std::string(who) + ": at most 32 batch columns per launch");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return api_fail(UNETMI_E_HIP, std::string(who) + ": cannot query the device");
    // placement assumes 8 XCDs of 32 compute units: hidden size 640 takes one workgroup on every one of them,
    // hidden size 300 (padded to 320) on 16 per XCD
    if (cus != kXcds * 32) return api_fail(UNETMI_E_UNSUPPORTED, std::string(who) + ": needs 8 XCDs of 32 compute units");
    return UNETMI_OK;
}

template <bool SAVE, int HP, bool F16 = false>
static int lstm_xcd_launch(const LstmXcdArgs& a, void* ws, hipStream_t s) {
    int per_cu = 0;
    const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_xcd_kernel<SAVE, HP, F16>, 256, 0);
    if (oe != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_recurrence_xcd occupancy query: ") + hipGetErrorString(oe));
    if (per_cu < 1) return api_fail(UNETMI_E_UNSUPPORTED, "lstm_recurrence_xcd: the kernel does not fit a compute unit");
    // tickets, abort word and the state granules start from zero: h_0 = 0 carrying step tag 0
    hipError_t e = hipMemsetAsync(ws, 0, unetmi_lstm_recurrence_xcd_workspace_bytes(), s);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_recurrence_xcd memset: ") + hipGetErrorString(e));
    hipLaunchKernelGGL((lstm_xcd_kernel<SAVE, HP, F16>), dim3(kXcds * (HP / kXU)), dim3(256), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("lstm_recurrence_xcd launch: ") + hipGetErrorString(e));
}

static int lstm_xcd_run(int T, int BP, int H, const float* wrec, const float* gx, float* hfull, int layout,
                        float* gates, float* cell, void* ws, size_t ws_bytes, void* stream) {
    if (T < 0 || BP < 1 || H < 1) return api_fail(UNETMI_E_BAD_DESC, "lstm_recurrence_xcd: bad sizes");
    if (layout < 0 || layout > 3) return api_fail(UNETMI_E_BAD_DESC, "lstm_recurrence_xcd: layout is 0 .. 3 (bit 0: channel-major, bit 1: fp16 operands)");
    const int channel_major = layout & 1;
    const bool f16 = (layout & 2) != 0;
    if (f16 && gates) return api_fail(UNETMI_E_UNSUPPORTED, "lstm_recurrence_xcd: the training form keeps fp32 operands");
    const int ok = lstm_xcd_device_ok("lstm_recurrence_xcd", H, BP);
    if (ok != UNETMI_OK) return ok;
    if (T == 0) return UNETMI_OK;
    if (!wrec || !gx || !hfull || !ws) return api_fail(UNETMI_E_BAD_ARG, "lstm_recurrence_xcd: null pointer");
    if (ws_bytes < unetmi_lstm_recurrence_xcd_workspace_bytes())
        return api_fail(UNETMI_E_BAD_ARG, "lstm_recurrence_xcd: workspace too small");
    LstmXcdArgs a{};
    a.wrec = wrec; a.gx = gx; a.hfull = hfull; a.T = T; a.BP = BP; a.H = H;
    a.gates = gates; a.cell = cell;
    if (channel_major) {        // [rows][T+1][BP]: slot 0 of both is the state before the first step
        a.gx_step = BP; a.gx_row = (long long)(T + 1) * BP; a.gx_first = BP;
        a.h_step = BP; a.h_row = (long long)(T + 1) * BP;
    } else {
        a.gx_step = (long long)4 * H * BP; a.gx_row = BP; a.gx_first = 0;
        a.h_step = (long long)((H + 15) / 16 * 16) * BP; a.h_row = BP;    // a step-major state slot is padded to 16 rows
    }
    a.ctl = static_cast<unsigned*>(ws);
    a.hx = reinterpret_cast<unsigned long long*>(static_cast<unsigned*>(ws) + 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (f16) return H == 640 ? lstm_xcd_launch<false, 640, true>(a, ws, s) : lstm_xcd_launch<false, 320, true>(a, ws, s);
    if (H == 640) return gates ? lstm_xcd_launch<true, 640>(a, ws, s) : lstm_xcd_launch<false, 640>(a, ws, s);
    return gates ? lstm_xcd_launch<true, 320>(a, ws, s) : lstm_xcd_launch<false, 320>(a, ws, s);
}

extern "C" size_t unetmi_lstm_bptt_xcd_workspace_bytes(void) {
    return 64 * sizeof(unsigned) + (size_t)kXcds * 2 * 4 * kXHMax * kXCols * sizeof(unsigned long long);
}

template <int HP>
static int lstm_bptt_launch(const LstmXcdBwdArgs& a, void* ws, hipStream_t s) {
    int per_cu = 0;
    const hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_xcd_bwd_kernel<HP>, 256, 0);
    if (oe != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_bptt_xcd occupancy query: ") + hipGetErrorString(oe));
    if (per_cu < 1) return api_fail(UNETMI_E_UNSUPPORTED, "lstm_bptt_xcd: the kernel does not fit a compute unit");
    // tickets, abort word and the granules start from zero: G_T = 0 carrying tag 0
    hipError_t e = hipMemsetAsync(ws, 0, unetmi_lstm_bptt_xcd_workspace_bytes(), s);
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("lstm_bptt_xcd memset: ") + hipGetErrorString(e));
    hipLaunchKernelGGL((lstm_xcd_bwd_kernel<HP>), dim3(kXcds * (HP / kXU)), dim3(256), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("lstm_bptt_xcd launch: ") + hipGetErrorString(e));
}

extern "C" int unetmi_lstm_bptt_xcd(int T, int BP, int H, const float* wrec, float* gates, const float* cell,
                                    const float* dhfull, void* ws, size_t ws_bytes, void* stream) {
    if (T < 0 || BP < 1 || H < 1) return api_fail(UNETMI_E_BAD_DESC, "lstm_bptt_xcd: bad sizes");
    const int ok = lstm_xcd_device_ok("lstm_bptt_xcd", H, BP);
    if (ok != UNETMI_OK) return ok;
    if (T == 0) return UNETMI_OK;
    if (!wrec || !gates || !cell || !dhfull || !ws) return api_fail(UNETMI_E_BAD_ARG, "lstm_bptt_xcd: null pointer");
    if (ws_bytes < unetmi_lstm_bptt_xcd_workspace_bytes()) return api_fail(UNETMI_E_BAD_ARG, "lstm_bptt_xcd: workspace too small");
    LstmXcdBwdArgs a{};
    a.wrec = wrec; a.gates = gates; a.cell = cell; a.dhfull = dhfull; a.T = T; a.BP = BP; a.H = H;
    a.ctl = static_cast<unsigned*>(ws);
    a.gxch = reinterpret_cast<unsigned long long*>(static_cast<unsigned*>(ws) + 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    return H == 640 ? lstm_bptt_launch<640>(a, ws, s) : lstm_bptt_launch<320>(a, ws, s);
}
