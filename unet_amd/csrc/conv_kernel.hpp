// Kernel template of the implicit-GEMM convolution (see conv_mfma.hip for the design
// notes).  Included by the instantiation units conv_inst_*.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_plan.hpp"
#include "io_types.hpp"

namespace unetmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* in0; const float* in1; const float* wpacked;
    const float* bias; const float* scale; const float* shift; const float* act_param;
    const float* gate; const float* residual; float* out;
    int N, C0, C1, Ctot, H, W, K, OH, OW;
    int in_mode, act, act_first, act_pc;
    int h_nphase, h_in_step, h_tap_step;
    int h_ntap[kMaxPhase], h_base[kMaxPhase], h_nout[kMaxPhase];
    int w_nphase, w_in_step, w_tap_step;
    int w_ntap[kMaxPhase], w_base[kMaxPhase], w_nout[kMaxPhase];
    int w_colmin;
    int ci_chunk, nchunk, nchunk0, NB, nco_blk, TI;
    int nrows, ncols, plane, in_tile_floats, w_tile_floats;
    int wtaps[kMaxPhase];
    unsigned long long w_phase_off[kMaxPhase];
    int nit[kMaxPhase], njt;
    unsigned blk_end[kMaxPhase];      // exclusive prefix end of each row phase's blocks
    unsigned magic_ncols, magic_nrows;  // ceil(2^32 / d) for division of small numbers
    int xcd_remap;                      // 1: one XCD walks a contiguous run of tiles
    int bf16;                           // activations (in0, in1, residual, out) stored as bfloat16
    double* stats;                      // training: [K][stats_S][2] per-tile (sum z, sum z^2) of the output, or null
    int stats_S;                        // tiles per output-channel block = partials per channel
#ifdef UNETMI_ABLATE
    int ablate;                        // developer probe only (tools/conv_probe.hip)
    unsigned long long* stamps;
#endif
};

#ifdef UNETMI_ABLATE
extern int g_ablate;
extern unsigned long long* g_stamps;   // [8] accumulated cycles per phase + block count (probe only)
#define STAMP(i) do { if (p.stamps) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    if (threadIdx.x == 0) atomicAdd(&p.stamps[i], t_ - t_last); t_last = t_; } } while (0)
#define ABLATE(bit) (p.ablate & (bit))
#else
#define ABLATE(bit) false
#define STAMP(i) do {} while (0)
#endif

// The activation is a compile-time parameter of the epilogue body (the run-time code picks the body once):
// a switch inlined at every result makes the kernel tens of thousands of instructions long and scatters
// the executed path over all of them.
template <int ACT>
__device__ __forceinline__ float apply_act(float v, float slope) {
    if constexpr (ACT == UNETMI_ACT_LEAKY)         return v >= 0.f ? v : v * slope;
    else if constexpr (ACT == UNETMI_ACT_RELU)     return fmaxf(v, 0.f);
    else if constexpr (ACT == UNETMI_ACT_SILU)     return v * fast_sigmoid(v);
    else if constexpr (ACT == UNETMI_ACT_SOFTPLUS) return fast_softplus(v);
    else if constexpr (ACT == UNETMI_ACT_SIGMOID)  return fast_sigmoid(v);
    else if constexpr (ACT == UNETMI_ACT_TANH)     return fast_tanh(v);
    else if constexpr (ACT == UNETMI_ACT_GELU)     return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    else                                           return v;
}

// ---------------------------------------------------------------------------------
// The MFMA loop.  Measured on gfx950: a v_mov into a register that an MFMA reads
// (the usual "next = load; cur = next" software pipeline) costs ~20 % of the
// matrix rate, and so does bunching the LDS reads after the last MFMA of a step.
// So operands live in two fixed register sets that alternate per tap (no copies),
// each tap's LDS reads are issued between the MFMAs of the previous tap, and the
// tap loop is unrolled so that set indices and LDS offsets are compile-time.
//
// A "row" is NTW taps at consecutive columns of one (channel pair, tap row); its
// reads are base-register + immediate.  NTW = 0 selects the generic form, where a
// row is a single tap and the walk also steps the column at run time.
// ---------------------------------------------------------------------------------
struct NoSide { __device__ __forceinline__ void operator()() const {} };

template <int TPG, int NTW, int ROWS, int NB, class Side>
__device__ __forceinline__ void mfma_rows(f32x16 (&acc)[TPG], float (&a)[2], float (&b)[2][TPG],
                                          const float* wp, const float* const (&rows)[ROWS + 1][TPG],
                                          Side& side) {
    constexpr int NT = ROWS * NTW;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int cur = i & 1, nxt = cur ^ 1;
        const int nr = (i + 1) / NTW, nc = (i + 1) % NTW;   // row / column of the NEXT tap
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], b[cur][t], acc[t], 0, 0, 0);
            if (t == 0) {
                a[nxt] = wp[(i + 1) * 2 * NB];
                b[nxt][0] = rows[nr][0][nc];
                if (TPG == 2) b[nxt][1] = rows[nr][1][nc];
            }
            if (TPG == 4 && t == 1) { b[nxt][1] = rows[nr][1][nc]; b[nxt][2] = rows[nr][2][nc]; }
            if (TPG == 4 && t == 2) b[nxt][3] = rows[nr][3][nc];
            if (TPG > 4 && t == 1) { b[nxt][1] = rows[nr][1][nc]; b[nxt][2] = rows[nr][2][nc]; }
            if (TPG > 4 && t >= 2 && t <= TPG - 2) b[nxt][t + 1] = rows[nr][t + 1][nc];
            __builtin_amdgcn_sched_barrier(0);
        }
        side();      // one piece of staging work per tap, issued in the shadow of the MFMAs above
#ifdef UNETMI_SIDE2
        side();      // probe: two per tap, so a chunk's DMA is all issued in the first half of the tap walk
#endif
    }
}

template <int TPG, int NTW, int NB, class Side>
__device__ __forceinline__ void group_mfma(f32x16 (&acc)[TPG], const float* s_in, const float* wp,
                                           const int (&ib)[TPG], int cpc, int nth, int ntw_rt,
                                           const ConvArgs& p, Side& side) {
    constexpr bool GENERIC = NTW == 0;
    constexpr int NTWE = GENERIC ? 1 : NTW;
    const int R = ABLATE(2) ? 0 : (GENERIC ? cpc * nth * ntw_rt : cpc * nth);
    if (R == 0) return;
    const int row_stride = p.h_tap_step * p.ncols;
    const int col_step = p.w_tap_step;
    const int row_wrap = row_stride - ntw_rt * col_step;       // generic: end of a tap row
    const int pair_wrap = 2 * p.plane - nth * row_stride;      // end of a channel pair
    int tw = 0, th = 0;
    auto advance = [&]() -> int {                              // offset from this row to the next
        int d;
        if (GENERIC) {
            ++tw; d = col_step;
            if (tw == ntw_rt) { tw = 0; d += row_wrap; ++th; if (th == nth) { th = 0; d += pair_wrap; } }
        } else {
            ++th; d = row_stride;
            if (th == nth) { th = 0; d += pair_wrap; }
        }
        return d;
    };
    // rows per unrolled block: a one-tap row (one-column kernels, the generic walk) is only TPG MFMAs, so four
    // of them share one pass of the loop's scalar bookkeeping; wider rows go two at a time
    constexpr int RW = NTWE == 1 ? 4 : 2;
    const float* rp[RW + 1][TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) rp[0][t] = s_in + ib[t];
    float a[2], b[2][TPG];
    a[0] = wp[0];
#pragma unroll
    for (int t = 0; t < TPG; ++t) b[0][t] = rp[0][t][0];
    int r = 0;
    for (; r + RW <= R; r += RW) {
#pragma unroll
        for (int q = 1; q <= RW; ++q) {
            const int d = (q < RW || r + RW < R) ? advance() : 0;      // past the end: re-read the last row
#pragma unroll
            for (int t = 0; t < TPG; ++t) rp[q][t] = rp[q - 1][t] + d;
        }
        mfma_rows<TPG, NTWE, RW, NB>(acc, a, b, wp, rp, side);
        wp += RW * NTWE * 2 * NB;
#pragma unroll
        for (int t = 0; t < TPG; ++t) rp[0][t] = rp[RW][t];
    }
    if (RW == 4) {
        for (; r + 2 <= R; r += 2) {
            const float* two[3][TPG];
            const int d1 = advance();
            const int d2 = (r + 2 < R) ? advance() : 0;
#pragma unroll
            for (int t = 0; t < TPG; ++t) { two[0][t] = rp[0][t]; two[1][t] = rp[0][t] + d1; two[2][t] = two[1][t] + d2; }
            mfma_rows<TPG, NTWE, 2, NB>(acc, a, b, wp, two, side);
            wp += 2 * NTWE * 2 * NB;
#pragma unroll
            for (int t = 0; t < TPG; ++t) rp[0][t] = two[2][t];
        }
    }
    if (r < R) {
        const float* last[2][TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) { last[0][t] = rp[0][t]; last[1][t] = rp[0][t]; }
        mfma_rows<TPG, NTWE, 1, NB>(acc, a, b, wp, last, side);
    }
}


// Two column phases (transposed conv with time stride 2) walked together: per tap row,
// the NTW0 taps of phase 0 then the NTW1 taps of phase 1, each on its own accumulators.
// One loop and one operand pipeline instead of two half-length ones.
template <int TPG, int NTW0, int NTW1, int ROWS, int NB, class Side>
__device__ __forceinline__ void mfma_rows2(f32x16 (&acc0)[TPG], f32x16 (&acc1)[TPG], float (&a)[2],
                                           float (&b)[2][TPG], const float* wp0, const float* wp1,
                                           const float* const (&rows0)[ROWS + 1][TPG],
                                           const float* const (&rows1)[ROWS + 1][TPG], Side& side) {
    constexpr int PR = NTW0 + NTW1;
    constexpr int NT = ROWS * PR;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int cur = i & 1, nxt = cur ^ 1;
        const bool g = (i % PR) >= NTW0;
        const int nrow = (i + 1) / PR, nk = (i + 1) % PR;
        const bool ng = nk >= NTW0;
        const int nc = ng ? nk - NTW0 : nk;
        const float* wnext = ng ? wp1 + (nrow * NTW1 + nc) * 2 * NB : wp0 + (nrow * NTW0 + nc) * 2 * NB;
        f32x16 (&acc)[TPG] = g ? acc1 : acc0;
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur], b[cur][t], acc[t], 0, 0, 0);
            if (t == 0) {
                a[nxt] = wnext[0];
                b[nxt][0] = (ng ? rows1 : rows0)[nrow][0][nc];
                if (TPG == 2) b[nxt][1] = (ng ? rows1 : rows0)[nrow][1][nc];
            }
            if (TPG == 4 && t == 1) {
                b[nxt][1] = (ng ? rows1 : rows0)[nrow][1][nc];
                b[nxt][2] = (ng ? rows1 : rows0)[nrow][2][nc];
            }
            if (TPG == 4 && t == 2) b[nxt][3] = (ng ? rows1 : rows0)[nrow][3][nc];
            if (TPG > 4 && t == 1) {
                b[nxt][1] = (ng ? rows1 : rows0)[nrow][1][nc];
                b[nxt][2] = (ng ? rows1 : rows0)[nrow][2][nc];
            }
            if (TPG > 4 && t >= 2 && t <= TPG - 2) b[nxt][t + 1] = (ng ? rows1 : rows0)[nrow][t + 1][nc];
            __builtin_amdgcn_sched_barrier(0);
        }
        side();
#ifdef UNETMI_SIDE2
        side();
#endif
    }
}

template <int TPG, int NTW0, int NTW1, int NB, class Side>
__device__ __forceinline__ void group_mfma2(f32x16 (&acc0)[TPG], f32x16 (&acc1)[TPG], const float* s_in,
                                            const float* wp0, const int (&ib0)[TPG], int phase_shift,
                                            int cpc, int nth, const ConvArgs& p, Side& side) {
    const int R = ABLATE(2) ? 0 : cpc * nth;
    if (R == 0) return;
    const float* wp1 = wp0 + R * NTW0 * 2 * NB;               // phase 1's weights follow phase 0's
    const int row_stride = p.h_tap_step * p.ncols;
    const int pair_wrap = 2 * p.plane - nth * row_stride;
    int th = 0;
    auto advance = [&]() -> int {
        ++th;
        int d = row_stride;
        if (th == nth) { th = 0; d += pair_wrap; }
        return d;
    };
    const float* rp0[3][TPG];
    const float* rp1[3][TPG];
#pragma unroll
    for (int t = 0; t < TPG; ++t) { rp0[0][t] = s_in + ib0[t]; rp1[0][t] = rp0[0][t] + phase_shift; }
    float a[2], b[2][TPG];
    a[0] = wp0[0];
#pragma unroll
    for (int t = 0; t < TPG; ++t) b[0][t] = rp0[0][t][0];
    int r = 0;
    for (; r + 2 <= R; r += 2) {
        const int d1 = advance();
        const int d2 = (r + 2 < R) ? advance() : 0;
#pragma unroll
        for (int t = 0; t < TPG; ++t) {
            rp0[1][t] = rp0[0][t] + d1; rp0[2][t] = rp0[1][t] + d2;
            rp1[1][t] = rp1[0][t] + d1; rp1[2][t] = rp1[1][t] + d2;
        }
        mfma_rows2<TPG, NTW0, NTW1, 2, NB>(acc0, acc1, a, b, wp0, wp1, rp0, rp1, side);
        wp0 += 2 * NTW0 * 2 * NB;
        wp1 += 2 * NTW1 * 2 * NB;
#pragma unroll
        for (int t = 0; t < TPG; ++t) { rp0[0][t] = rp0[2][t]; rp1[0][t] = rp1[2][t]; }
    }
    if (r < R) {
        const float* l0[2][TPG];
        const float* l1[2][TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) { l0[0][t] = l0[1][t] = rp0[0][t]; l1[0][t] = l1[1][t] = rp1[0][t]; }
        mfma_rows2<TPG, NTW0, NTW1, 1, NB>(acc0, acc1, a, b, wp0, wp1, l0, l1, side);
    }
}

// Training: per-channel (sum z, sum z^2) of this workgroup's output tile for BatchNorm's batch statistics, taken from
// the accumulators so that the statistics need no pass over the tensor.  z = (acc + bias) * scale + shift (the layer
// runs without activation, gate or added tensor in this mode).  A lane adds its pixel's values over the wave's rows
// and column phases, the 32 lanes of a half-wave (= 32 pixels of the same channels) are added by a fixed butterfly,
// the WM waves that share channels meet in LDS (the tile buffers are free after the matrix loop) and are added in
// wave order in double: the result does not depend on launch order.  Slot tile_lin of every channel of this
// workgroup's block is written by exactly this workgroup.
template <int CTRL>
__device__ __forceinline__ float stats_dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, moved);
}

template <int NG, int TPG>
__device__ __forceinline__ void conv_tile_stats(const f32x16 (&acc)[NG][TPG], const ConvArgs& p, const float* s_par,
                                                float* s_red, int WM, int cb, int a, int i0, int j0, int wm, int wn,
                                                int half, int j, unsigned tile_lin) {
    __syncthreads();                                  // every wave has left the matrix loop: s_red may be reused
    const int col_base = wn * 32 + 4 * half;
    const int jj = j0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = col_base + (r & 3) + 8 * (r >> 2);
        const float b = s_par[cl], sc = s_par[p.NB + cl], sf = s_par[2 * p.NB + cl];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool col_ok = jj < p.w_nout[g];
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                const bool ok = col_ok && (i0 + wm * TPG + t) < p.h_nout[a];
                const float v = ok ? (acc[g][t][r] + b) * sc + sf : 0.f;
                s1 += v;
                s2 += v * v;
            }
        }
        s1 = stats_dpp_add<0xB1>(s1); s2 = stats_dpp_add<0xB1>(s2);        // quad_perm [1,0,3,2]
        s1 = stats_dpp_add<0x4E>(s1); s2 = stats_dpp_add<0x4E>(s2);        // quad_perm [2,3,0,1]
        s1 = stats_dpp_add<0x141>(s1); s2 = stats_dpp_add<0x141>(s2);      // row_half_mirror
        s1 = stats_dpp_add<0x140>(s1); s2 = stats_dpp_add<0x140>(s2);      // row_mirror
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        if (j == 0) { s_red[(wm * p.NB + cl) * 2] = s1; s_red[(wm * p.NB + cl) * 2 + 1] = s2; }
    }
    __syncthreads();
    const int tid = threadIdx.x;
    if (tid < p.NB && cb * p.NB + tid < p.K) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < WM; ++w) { t1 += (double)s_red[(w * p.NB + tid) * 2]; t2 += (double)s_red[(w * p.NB + tid) * 2 + 1]; }
        double* dst = p.stats + ((size_t)(cb * p.NB + tid) * p.stats_S + tile_lin) * 2;
        dst[0] = t1;
        dst[1] = t2;
    }
    __syncthreads();                                  // s_par is read again by the epilogue; s_red is not touched after this
}

// Fused epilogue: bias, BatchNorm affine and activation (either order), gate, residual,
// store.  Lane = one time column, register = one output channel (MFMA C/D layout).
template <int NG, int TPG, bool BF16, int ACT>
__device__ __forceinline__ void conv_epilogue_body(const f32x16 (&acc)[NG][TPG], const ConvArgs& p,
                                                   const float* s_par, int n, int cb, int a, int i0, int j0,
                                                   int wm, int wn, int half, int j) {
    const int col_base = wn * 32 + 4 * half;          // this lane's first channel inside the block
    const size_t OHW = (size_t)p.OH * p.OW;
    float e_b[16], e_sc[16], e_sf[16], e_sl[16], e_g[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cl = col_base + (r & 3) + 8 * (r >> 2);
        e_b[r] = s_par[cl]; e_sc[r] = s_par[p.NB + cl]; e_sf[r] = s_par[2 * p.NB + cl];
        e_sl[r] = s_par[3 * p.NB + cl]; e_g[r] = s_par[4 * p.NB + cl];
    }
    const int jj = j0 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = cb * p.NB + col_base + (r & 3) + 8 * (r >> 2);
        if (co >= p.K) continue;
        const size_t chan = ((size_t)n * p.K + co) * OHW;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ow = jj * p.w_nphase + g;
            const bool col_ok = jj < p.w_nout[g];
#pragma unroll
            for (int t = 0; t < TPG; ++t) {
                const int i = i0 + wm * TPG + t;
                if (i >= p.h_nout[a] || !col_ok) continue;
                const int oh = i * p.h_nphase + a;
                float v = acc[g][t][r] + e_b[r];
                if (p.act_first) v = apply_act<ACT>(v, e_sl[r]) * e_sc[r] + e_sf[r];
                else             v = apply_act<ACT>(v * e_sc[r] + e_sf[r], e_sl[r]);
                v *= e_g[r];
                const size_t o = chan + (size_t)oh * p.OW + ow;
                if (p.residual) v += ld_act<BF16>(p.residual, o);
                if (!ABLATE(4) || v == 123.456f) st_act<BF16>(p.out, o, v);
            }
        }
    }
}

// The same epilogue for kernels with ONE column phase (NG == 1), without branches or 64-bit address arithmetic:
// the output (and the residual) of sample n is addressed through a buffer descriptor over that sample, the
// lane-dependent part of the address (column, and which half of the channel group the lane holds) is ONE byte
// offset per lane for all its results, and the part that changes from result to result (channel, row) is
// wave-uniform and travels as the instruction's scalar offset.  A lane whose column or channel is outside the
// tensor carries an offset beyond the descriptor's range, so its loads return zero and its stores are dropped.
// All residual values of a row group are requested before the first one is used.
template <int TPG, bool BF16, int ACT>
__device__ __forceinline__ void conv_epilogue_body_ng1(const f32x16 (&acc)[1][TPG], const ConvArgs& p,
                                                       const float* s_par, int n, int cb, int a, int i0, int j0,
                                                       int wm, int wn, int half, int j) {
    constexpr int EB = BF16 ? 2 : 4;
    constexpr int kDrop = (int)0x80000000;
    const int col_base = wn * 32 + 4 * half;
    const int OHW = p.OH * p.OW;
    const int sample_bytes = p.K * OHW * EB;                   // < 2^31: checked when the layer is planned
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(
        static_cast<char*>(static_cast<void*>(p.out)) + (size_t)n * sample_bytes, 0, sample_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(static_cast<const char*>(static_cast<const void*>(p.residual ? p.residual : p.out))) +
            (size_t)n * sample_bytes, 0, p.residual ? sample_bytes : 0, 0x00020000);
    const int jj = j0 + j;
    const int co0 = cb * p.NB + col_base;                      // the lane's first channel
    // byte offset of (channel 4*half of this wave's 32, row 0, this lane's column)
    const int lane_off = jj < p.w_nout[0] ? (4 * half * OHW + jj * p.w_nphase) * EB : kDrop;
    const int chan0 = cb * p.NB + wn * 32;                     // uniform: first channel of this wave's 32
#pragma unroll
    for (int t = 0; t < TPG; ++t) {
        const int i = i0 + wm * TPG + t;
        if (i >= p.h_nout[a]) continue;                        // wave-uniform
        const int oh = i * p.h_nphase + a;
        const int row_off = (chan0 * p.OH + oh) * p.OW * EB;   // uniform byte offset of (channel chan0, row oh)
        int voff[16];
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);              // channel of register r, relative to the lane's first
            voff[r] = (co0 + c < p.K) ? lane_off : kDrop;
            res[r] = 0.f;
        }
        if (p.residual) {                                      // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r)
                res[r] = buf_ld_act<BF16>(rs_res, voff[r], row_off + ((r & 3) + 8 * (r >> 2)) * OHW * EB);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            const int cl = col_base + c;
            float v = acc[0][t][r] + s_par[cl];
            const float sc = s_par[p.NB + cl], sf = s_par[2 * p.NB + cl], sl = s_par[3 * p.NB + cl];
            if (p.act_first) v = apply_act<ACT>(v, sl) * sc + sf;
            else             v = apply_act<ACT>(v * sc + sf, sl);
            v = v * s_par[4 * p.NB + cl] + res[r];
            if (!ABLATE(4) || v == 123.456f) buf_st_act<BF16>(v, rs_out, voff[r], row_off + c * OHW * EB);
        }
    }
}

template <int NG, int TPG, bool BF16 = false>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[NG][TPG], const ConvArgs& p,
                                              const float* s_par, int n, int cb, int a, int i0, int j0,
                                              int wm, int wn, int half, int j) {
    if constexpr (NG == 1) {
        switch (p.act) {
#define UNETMI_EPI1(A_) \
        case A_: conv_epilogue_body_ng1<TPG, BF16, A_>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j); return;
            UNETMI_EPI1(UNETMI_ACT_LEAKY) UNETMI_EPI1(UNETMI_ACT_RELU) UNETMI_EPI1(UNETMI_ACT_SILU)
            UNETMI_EPI1(UNETMI_ACT_SOFTPLUS) UNETMI_EPI1(UNETMI_ACT_SIGMOID) UNETMI_EPI1(UNETMI_ACT_TANH)
            UNETMI_EPI1(UNETMI_ACT_GELU)
#undef UNETMI_EPI1
            default: conv_epilogue_body_ng1<TPG, BF16, UNETMI_ACT_NONE>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j);
        }
        return;
    }
    switch (p.act) {
#define UNETMI_EPI(A_) \
    case A_: conv_epilogue_body<NG, TPG, BF16, A_>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j); break;
        UNETMI_EPI(UNETMI_ACT_LEAKY) UNETMI_EPI(UNETMI_ACT_RELU) UNETMI_EPI(UNETMI_ACT_SILU)
        UNETMI_EPI(UNETMI_ACT_SOFTPLUS) UNETMI_EPI(UNETMI_ACT_SIGMOID) UNETMI_EPI(UNETMI_ACT_TANH)
        UNETMI_EPI(UNETMI_ACT_GELU)
#undef UNETMI_EPI
        default: conv_epilogue_body<NG, TPG, BF16, UNETMI_ACT_NONE>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j);
    }
}

template <int NG, int TPG, int WM, int WN, bool TWO, int NTW0, int NTW1, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvArgs p) {
    constexpr int EB = BF16 ? 2 : 4;              // bytes of a stored activation
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + p.in_tile_floats;
    float* s_par = s_w + p.w_tile_floats;      // [5][NB]: bias, scale, shift, slope, gate

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef UNETMI_ABLATE
    unsigned long long t_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_last) :: "memory");
#endif
    const int half = lane >> 5, j = lane & 31;

    // ---- which tile is this workgroup's -----------------------------------
    // Workgroups are dealt round-robin over the 8 XCDs (private L2 each).  Re-number
    // them so that one XCD walks a CONTIGUOUS run of tiles: neighbouring tiles share
    // input halo rows/columns and the cache lines their output rows straddle, which
    // then meet in one L2 instead of being written back as partial lines.
    unsigned bid = blockIdx.x;
    if (p.xcd_remap) {
        const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7u, q = nwg >> 3, r = nwg & 7u;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const unsigned bid_all = bid;
    int a = 0;
    unsigned start = 0;
#pragma unroll
    for (int q = 0; q < kMaxPhase - 1; ++q)
        if (q + 1 < p.h_nphase && bid >= p.blk_end[q]) { a = q + 1; start = p.blk_end[q]; }
    bid -= start;
    const int cb = bid % p.nco_blk; bid /= p.nco_blk;
    const int jt = bid % p.njt;     bid /= p.njt;
    const int it = bid % p.nit[a];  bid /= p.nit[a];
    const int n = bid;
    const int i0 = it * p.TI, j0 = jt * kTileJ;
    const int nth = p.h_ntap[a];
    const int row0 = i0 * p.h_in_step + p.h_base[a];
    const int col0 = j0 * p.w_in_step + p.w_colmin;
    const int cpc = p.ci_chunk >> 1;
    const int chunk_w_floats = cpc * p.wtaps[a] * 2 * p.NB;
    const int HW = p.H * p.W;

    f32x16 acc[NG][TPG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][t][r] = 0.f;

    // per-lane LDS offsets of the B operand (input patch) for each accumulator tile
    int ib[NG][TPG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < TPG; ++t)
            ib[g][t] = half * p.plane + ((wm * TPG + t) * p.h_in_step) * p.ncols +
                       j * p.w_in_step + (p.w_base[g] - p.w_colmin);
    const int wlane = half * p.NB + wn * 32 + j;

    // ---- per-thread prefetch slots: element e = s*256 + tid of the chunk image ----
    // Every load below is an unconditional buffer load: lanes that fall outside
    // the image, past the last channel, or in an unused slot carry an offset
    // beyond the descriptor's range and get 0 back without touching memory.  No
    // branch surrounds a load, so all of a chunk's loads are in flight together.
    const int in_elems = p.ci_chunk * p.plane;
    const int nin_slots = (in_elems + 255) >> 8;
    const int n4 = chunk_w_floats >> 2;
    const int nw_slots = (n4 + 255) >> 8;
    constexpr int kOutside = (int)0x80000000;
    int voff[kMaxInSlots];             // byte offset inside the chunk's channel block
#pragma unroll
    for (int s = 0; s < kMaxInSlots; ++s) {
        const int e = s * 256 + tid;
        const int rr = __umulhi((unsigned)e, p.magic_ncols);
        const int c = e - rr * p.ncols;
        const int cl = (p.nrows == 1) ? rr : (int)__umulhi((unsigned)rr, p.magic_nrows);
        const int r = rr - cl * p.nrows;
        const int grow = row0 + r, gcol = col0 + c;
        const bool ok = e < in_elems && (unsigned)grow < (unsigned)p.H && (unsigned)gcol < (unsigned)p.W;
        voff[s] = ok ? (cl * HW + grow * p.W + gcol) * EB : kOutside;
    }
    float rin[kMaxInSlots];
    float rin2[TWO ? kMaxInSlots : 1];
    f32x4 rw[kMaxW4Slots];
    const float* wsrc_base = p.wpacked + p.w_phase_off[a] + (size_t)cb * p.nchunk * chunk_w_floats;

    auto prefetch = [&](int chunk) {
        const bool live = chunk < p.nchunk && !(ABLATE(8) && chunk > 0);   // else: every load out of range
        // source 0: the tensor this chunk streams from (a concatenation is chunked per source);
        // source 1: the addend of a summed input
        const bool second = chunk >= p.nchunk0;
        const int ci_base = (second ? chunk - p.nchunk0 : chunk) * p.ci_chunk;
        const int csrc = second ? p.C1 : p.C0;
        const int ch0 = max(min(p.ci_chunk, csrc - ci_base), 0);
        const long long first0 = (long long)n * csrc + ci_base;
        const int ch1 = (TWO && p.in_mode == UNETMI_IN_ADD) ? ch0 : 0;
        const long long first1 = first0;
        const int shift1 = 0;
        const char* base0 = reinterpret_cast<const char*>(second ? p.in1 : p.in0);
        const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(base0 + first0 * HW * EB), 0, live ? ch0 * HW * EB : 0, 0x00020000);
#pragma unroll
        for (int s = 0; s < kMaxInSlots; ++s) rin[s] = buf_ld_act<BF16>(rs0, voff[s], 0);
        if (TWO) {
            const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(p.in1) + first1 * HW * EB), 0,
                live ? ch1 * HW * EB : 0, 0x00020000);
#pragma unroll
            for (int s = 0; s < kMaxInSlots; ++s) rin2[s] = buf_ld_act<BF16>(rs1, voff[s] + shift1, 0);
        }
        const auto rsw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(wsrc_base + (size_t)(live ? chunk : 0) * chunk_w_floats), 0,
            live ? chunk_w_floats * 4 : 0, 0x00020000);
#pragma unroll
        for (int s = 0; s < kMaxW4Slots; ++s)
            rw[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rsw, (s * 256 + tid) * 16, 0, 0));
    };

    // Epilogue parameters of this workgroup's output channels: fetched now, parked in
    // LDS with the first chunk, read back in one batch by the epilogue (a global load
    // per output register there would expose one memory round trip per register).
    float par[5] = {0.f, 1.f, 0.f, 0.f, 1.f};
    if (tid < p.NB) {
        const int co = cb * p.NB + tid;
        if (co < p.K) {
            par[0] = p.bias[co]; par[1] = p.scale[co]; par[2] = p.shift[co];
            if (p.act_param) par[3] = p.act_param[p.act_pc ? co : 0];
            if (p.gate) par[4] = p.gate[(size_t)n * p.K + co];
        }
    }
    STAMP(0);                      // prologue: tile decode, offsets
    prefetch(0);
    STAMP(1);                      // issue of the first prefetch
    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        __syncthreads();   // every wave is done reading the previous chunk from LDS
        STAMP(2);                  // barrier after compute
        if (!(ABLATE(1) && chunk > 0)) {
#pragma unroll
        for (int s = 0; s < kMaxInSlots; ++s)
            if (s < nin_slots)     // LDS image is padded to whole slots: no per-lane guard
                s_in[s * 256 + tid] = TWO ? rin[s] + rin2[s] : rin[s];
#pragma unroll
        for (int s = 0; s < kMaxW4Slots; ++s)
            if (s < nw_slots) reinterpret_cast<f32x4*>(s_w)[s * 256 + tid] = rw[s];
        }
        if (chunk == 0 && tid < p.NB) {
#pragma unroll
            for (int q = 0; q < 5; ++q) s_par[q * p.NB + tid] = par[q];
        }
        STAMP(3);                  // wait for the prefetch + LDS writes
        __syncthreads();
        STAMP(4);                  // barrier after LDS writes
        prefetch(chunk + 1);       // in flight while this chunk computes
        STAMP(5);                  // issue of the next prefetch

        // ---- MFMA over (column phase; channel pair x tap row x tap col) -------------
        {
            const float* wp = s_w + wlane;
            NoSide none;
            group_mfma<TPG, NTW0, 32 * WN>(acc[0], s_in, wp, ib[0], cpc, nth, p.w_ntap[0], p, none);
            if (NG == 2) {
                wp += cpc * nth * p.w_ntap[0] * 2 * p.NB;
                group_mfma<TPG, NTW1, 32 * WN>(acc[NG - 1], s_in, wp, ib[NG - 1], cpc, nth, p.w_ntap[1], p, none);
            }
        }
    }

    STAMP(6);                      // MFMA loops (all chunks)
    if (p.stats)                   // workgroup-uniform
        conv_tile_stats<NG, TPG>(acc, p, s_par, smem, WM, cb, a, i0, j0, wm, wn, half, j, bid_all / p.nco_blk);
    // ---- epilogue ---------------------------------------------------------------
    conv_epilogue<NG, TPG, BF16>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j);
    STAMP(7);                      // epilogue
#ifdef UNETMI_ABLATE
    if (p.stamps && threadIdx.x == 0) atomicAdd(&p.stamps[8], 1ull);
#endif
}

template <int NG, int TPG, int WM, int WN, bool TWO, int NTW0, int NTW1, bool BF16 = false>
inline void launch_conv_variant(unsigned grid, size_t lds, hipStream_t stream, const ConvArgs& p) {
    hipLaunchKernelGGL((conv_mfma_kernel<NG, TPG, WM, WN, TWO, NTW0, NTW1, BF16>), dim3(grid), dim3(256), lds,
                       stream, p);
}

// Instantiation units.  Each returns false when it holds no kernel for the request.
bool launch_conv_staged(int ng, int wn, bool two, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p);
bool launch_conv_dma_generic(int ng, int wn, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p);
bool launch_conv_ng1(int wn, int ntw0, int tpg, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p);
bool launch_conv_ng2(int wn, int ntw0, int ntw1, int tpg, unsigned grid, size_t lds, hipStream_t s, const ConvArgs& p);

}  // namespace unetmi
