// Convolution / transposed convolution with bfloat16 OPERANDS on the matrix cores
// (v_mfma_f32_32x32x16_bf16, 16x the rate of the exact-fp32 MFMA) over a
// channel-blocked activation layout.
//
// Layout ("blocked"): [N][CB][H][W][8] bfloat16, CB = 2 * ceil(C / 16): eight consecutive channels
// of one pixel are one 16-byte record, channels are padded with zeros to a multiple of 16.  That
// is exactly what one lane of the bf16 MFMA consumes (8 consecutive K values), so a B operand is
// ONE ds_read_b128 and an input row reaches LDS by 16-byte LDS-DMA without passing registers.
//
// GEMM view: M = output channels (A = weights, 32 per tile, MT tiles per wave), N = 32 * NT pixels
// along time, K = (16-channel group, tap).  A workgroup is 8 waves = 8 output rows x 32 NT columns x
// all output channels.  One pipeline step = (channel group, weight half, tap row): its weights [tap col][MT][32][16]
// arrive two steps ahead in a 3-deep LDS ring, the group's input rows one group ahead (two when a
// tap row is a whole group), so HBM / L2 latency is covered by one or two steps of MFMA work and a
// step ends with one s_barrier and a counted vmcnt wait -- never a full drain.
//
// Strided and transposed layers use the same loop: every layer is a set of "phases", each a
// stride-1 walk over its own output sub-grid with  ih = i*sih + t*tdh + offh,  oh = i*soh + poh
// (plain conv: one phase, sih = stride, tdh = dilation; transposed: stride_h * stride_w phases with
// sih = 1, tdh = -1 and the taps of that phase only).
//
// Weights are bfloat16 (rounded to nearest-even by the packer); with hl = 2 the packer also stores
// the rounding remainder as a second bfloat16 image and the kernel issues a second MFMA with it,
// which reproduces fp32-weight arithmetic (bf16 x bf16 products are exact in fp32).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {
int api_fail(int code, const std::string& msg);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float b16_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* b16_lds_ptr;
typedef __bf16 b16_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b16_f32x2 __attribute__((ext_vector_type(2)));
typedef float b16_f32x4 __attribute__((ext_vector_type(4)));
typedef float b16_f32x4_u __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte access at element alignment

typedef _Float16 b16_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 b16_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned b16_pack2(float lo, float hi) {        // v_cvt_pk_bf16_f32: nearest-even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(b16_f32x2{lo, hi}, b16_bf16x2));
}
// two fp16 values, nearest-even.  A magnitude beyond the largest finite fp16 value (65504) becomes infinity, as the
// format defines, and turns into NaN in the next layer's products: a tensor that does not fit the format shows in
// the result instead of being clipped silently.
__device__ __forceinline__ unsigned b16_pack2_f16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(b16_f32x2{lo, hi}, b16_f16x2));
}
// the 16-bit storage format of a blocked tensor is a run-time property of the layer's OUTPUT
__device__ __forceinline__ unsigned b16_pack2_as(float lo, float hi, int f16) {
    return f16 ? b16_pack2_f16(lo, hi) : b16_pack2(lo, hi);
}
// ... and a compile-time property of its OPERANDS (both inputs and the weight image): same matrix rate either way
template <bool F16>
__device__ __forceinline__ b16_f32x16 b16_mfma(bf16x8 a, bf16x8 b, b16_f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(b16_f16x8, a), __builtin_bit_cast(b16_f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

#ifndef B16_ABLATE
#define B16_ABLATE 0      // build-time experiment switches (tools/b16_probe.py); 0 in the shipped library:
                          // 1 no MFMA loop, 2 no operand DMA after the prologue, 4 no epilogue, 8 no step barrier, 16 no LDS reads
#endif
constexpr int kB16MaxPhase = 4;
constexpr int kB16Waves = 8;          // waves (= output rows) per workgroup
constexpr int kB16MaxQ = 8;           // input DMA instructions per wave and group (64 KB image)

// Column quantities come in pairs: a transposed layer with time stride 2 runs its two column phases as the
// two column tiles of one workgroup ("pair" mode), so that the workgroup writes whole output lines; otherwise
// entry 1 repeats entry 0.
struct B16Phase {
    int nth, ntw[2];         // taps of this phase
    int offh, offw[2];       // ih = i*sih + t*tdh + offh
    int poh, pow_[2];        // oh = i*soh + poh
    int rows, cols[2];       // extent of this phase's output grid
    unsigned wg_end;         // workgroups up to and including this phase
    unsigned w_off;          // byte offset of the phase's weight image
    unsigned w_bytes;        // its size
};

struct B16Args {
    const char* in0; const char* in1; const char* w;
    const float* bias; const float* scale; const float* shift; const float* act_param; const float* gate;
    void* out;
    int N, H, W, OH, OW, K;  // K: output channels of ONE workgroup column (<= 128)
    int Kfull;               // output channels of the layer: gridDim.y chunks of 128 when it has more than 128
    unsigned w_chunk_bytes;  // size of one chunk's weight image
    int G0, G1;              // 16-channel groups of the two sources
    int sih, siw, tdh, tdw, soh, sow;
    int nphase, hl, din;     // weight halves (1 | 2); input prefetch distance in groups (1 | 2)
    int wring, allrows;      // weight ring depth (2 | 3); 1: one step holds every tap row of a group
    int pair;                // 1: column tiles = the two column phases
    int in_buf_bytes, w_buf_bytes;
    int act, act_first, act_pc, out_nchw_f32;
    int out_f16;             // blocked output stored as fp16 instead of bfloat16
    // fused 1x1 head: out = head_act(head_w [P][K] . layer_output + head_b) per pixel, written as fp32 instead of the
    // layer's own output (head_P = 0: no head).  head_il: [N][OH][OW][P] instead of [N][P][OH][OW].
    const float* residual;   // out_nchw_f32 only: fp32 [N,K,OH,OW] added to the result, or null
    const float* head_w; const float* head_b;
    int head_P, head_act, head_il;
    B16Phase ph[kB16MaxPhase];
};

constexpr int kB16MaxHead = 4;        // output channels of a fused head

// Activations.  LeakyReLU / PReLU / ReLU / none are one select with a per-channel slope (1 for none, 0 for
// ReLU) and are what the wide layers use.  The transcendental ones go through a run-time switch, which
// is only instantiated for layers of at most 32 output channels (the mask heads): inlined once per
// accumulator register of a wide tile that switch made the epilogue ~80,000 instructions of mostly
// skipped code, and the instruction-cache misses of the skips cost more than the whole matrix loop.
__device__ __forceinline__ float b16_act_general(float v, int act, float slope) {
    switch (act) {
        case UNETMI_ACT_SILU:     return v * fast_sigmoid(v);
        case UNETMI_ACT_SOFTPLUS: return fast_softplus(v);
        case UNETMI_ACT_SIGMOID:  return fast_sigmoid(v);
        case UNETMI_ACT_TANH:     return fast_tanh(v);
        case UNETMI_ACT_GELU:     return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        default:                  return v >= 0.f ? v : v * slope;
    }
}

// The squash of a fused head: the activations built from v_exp / v_log / v_rcp only (no GELU: erff is a long
// polynomial with branches).
__device__ __forceinline__ float b16_head_act(float v, int act) {
    switch (act) {
        case UNETMI_ACT_RELU:     return fmaxf(v, 0.f);
        case UNETMI_ACT_SILU:     return v * fast_sigmoid(v);
        case UNETMI_ACT_SOFTPLUS: return fast_softplus(v);
        case UNETMI_ACT_SIGMOID:  return fast_sigmoid(v);
        case UNETMI_ACT_TANH:     return fast_tanh(v);
        default:                  return v;
    }
}

// f(integral_constant<int, I>) for I = FROM .. TO-1 with I a compile-time value
template <int FROM, int TO, class F>
__device__ __forceinline__ void static_for_b16(F&& f) {
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        static_for_b16<FROM + 1, TO>(f);
    }
}

// LDS byte address of a pointer into the workgroup's shared memory
__device__ __forceinline__ int b16_lds_addr(const void* q) {
    return (int)(unsigned)(size_t)(__attribute__((address_space(3))) const char*)q;
}
// 16 bytes from LDS at (per-lane base) + (wave-uniform offset) + (immediate)
__device__ __forceinline__ bf16x8 b16_lds_ld(int lane_base, int uniform, int imm) {
    typedef const __attribute__((address_space(3))) bf16x8* lds_cv8;
    if (B16_ABLATE & 16) {                       // timing experiment: operands without LDS reads
        bf16x8 v;
        asm volatile("" : "=v"(v) : "s"(uniform));
        return v;
    }
    return *reinterpret_cast<lds_cv8>(static_cast<unsigned>(lane_base + uniform) + static_cast<unsigned>(imm));
}
// a wave-uniform value, kept in a scalar register and hidden from the optimizer's induction-variable rewriting
__device__ __forceinline__ int b16_keep_scalar(int v) {
    asm volatile("" : "+s"(v));
    return v;
}

// wait until at most n of this wave's vector-memory operations (LDS-DMA included) are outstanding
__device__ __forceinline__ void b16_wait_vm(int n) {
    switch (n) {
        case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1:  asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2:  asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3:  asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4:  asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5:  asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6:  asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7:  asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8:  asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9:  asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Bias, folded BatchNorm, activation and gate on the fp32 accumulators, then either 8-byte stores of
// four bfloat16 channels (lanes l and l+32 complete a pixel's 16-byte record, a wave writes 512
// contiguous bytes per instruction) or plain fp32 NCHW stores for the heads.
// FOLD: the prologue found the layer's per-channel arithmetic reducible to  act(acc * A + B)  with one slope for every
// channel (bias and BatchNorm folded into A, B; a non-negative gate folded too, since act(z) * g = act(z * g) for
// g >= 0 with a leaky / ReLU / identity activation): two 16-byte parameter reads per four channels instead of
// twenty 4-byte ones -- a workgroup's eight waves otherwise spend more LDS cycles on parameters than on a
// step's operands.  s_par rows then are A, B, -, slope, -.
template <int MT, int NT, bool GEN, bool PAIR, bool FOLD>
__device__ __forceinline__ void b16_epilogue(const b16_f32x16 (&acc)[MT][NT], const B16Args& p, const B16Phase& ph,
                                             const float* s_par, int n, int oh, int j0, int half, int j, int kb) {
    static_assert(!(GEN && FOLD), "the folded form has leaky / ReLU / identity activations only");
    // kb: first output channel of this workgroup's chunk (0 unless the layer has more than 128 outputs)
    const int CBo = 2 * ((p.Kfull + 15) >> 4);
    const int kblk = kb >> 3;
    const size_t OHW = (size_t)p.OH * p.OW;
    size_t pix[NT];
    bool ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int jc = PAIR ? j0 + j : j0 + nt * 32 + j;
        ok[nt] = jc < ph.cols[PAIR ? nt : 0];
        pix[nt] = (size_t)oh * p.OW + (size_t)(jc * p.sow + ph.pow_[PAIR ? nt : 0]);
    }
    const float slope0 = s_par[3 * MT * 32];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int blk = mt * 4 + q;
            const int m0 = blk * 8 + half * 4;
            float v[NT][4];
            if constexpr (FOLD) {
                const b16_f32x4 A = *reinterpret_cast<const b16_f32x4*>(s_par + m0);
                const b16_f32x4 B = *reinterpret_cast<const b16_f32x4*>(s_par + MT * 32 + m0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = fmaf(acc[mt][nt][q * 4 + r], A[r], B[r]);
                        v[nt][r] = x >= 0.f ? x : x * slope0;
                    }
            } else {
                float bi[4], sc[4], sf[4], sl[4], gt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bi[r] = s_par[m0 + r]; sc[r] = s_par[MT * 32 + m0 + r]; sf[r] = s_par[2 * MT * 32 + m0 + r];
                    sl[r] = s_par[3 * MT * 32 + m0 + r]; gt[r] = s_par[4 * MT * 32 + m0 + r];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[mt][nt][q * 4 + r] + bi[r];
                        if (p.act_first) x = (GEN ? b16_act_general(x, p.act, sl[r]) : (x >= 0.f ? x : x * sl[r])) * sc[r] + sf[r];
                        else {
                            x = x * sc[r] + sf[r];
                            x = GEN ? b16_act_general(x, p.act, sl[r]) : (x >= 0.f ? x : x * sl[r]);
                        }
                        v[nt][r] = x * gt[r];
                    }
            }
            if (NT == 2 && !PAIR && p.out_nchw_f32 && p.sow == 1) {
                // fp32 NCHW, two column tiles side by side: after the lane halves of the two tiles change places a
                // lane holds pixel j0 + lane of channel m0' + r (first register) and of m0' + 4 + r (second), so one
                // store instruction writes 64 consecutive pixels of one channel; the added tensor is read the same way,
                // all eight loads of a channel quad ahead of its stores
                const int jc = j0 + (half * 32 + j);
                const bool okc = jc < ph.cols[0];
                const size_t pixc = (size_t)oh * p.OW + (size_t)(jc + ph.pow_[0]);
                const int c0 = blk * 8;                          // channels c0 + r and c0 + 4 + r
                float* const out = static_cast<float*>(p.out);
                float add[2][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    add[0][r] = (p.residual && okc && c0 + r < p.K) ? p.residual[((size_t)n * p.Kfull + kb + c0 + r) * OHW + pixc] : 0.f;
                    add[1][r] = (p.residual && okc && c0 + 4 + r < p.K) ? p.residual[((size_t)n * p.Kfull + kb + c0 + 4 + r) * OHW + pixc] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned s0 = __builtin_bit_cast(unsigned, v[0][r]), s1 = __builtin_bit_cast(unsigned, v[1][r]);
                    swap_lane_halves(s0, s1);
                    if (okc && c0 + r < p.K)
                        out[((size_t)n * p.Kfull + kb + c0 + r) * OHW + pixc] = __builtin_bit_cast(float, s0) + add[0][r];
                    if (okc && c0 + 4 + r < p.K)
                        out[((size_t)n * p.Kfull + kb + c0 + 4 + r) * OHW + pixc] = __builtin_bit_cast(float, s1) + add[1][r];
                }
            } else if (p.out_nchw_f32) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (ok[nt] && m0 + r < p.K) {
                            const size_t o = ((size_t)n * p.Kfull + kb + m0 + r) * OHW + pix[nt];
                            static_cast<float*>(p.out)[o] = p.residual ? v[nt][r] + p.residual[o] : v[nt][r];
                        }
            } else if constexpr (NT == 2) {
                // lanes l and l+32 hold the two halves of a pixel's record in each column tile: swapping the upper
                // lanes of tile 0 with the lower lanes of tile 1 leaves every lane with one whole 16-byte record
                // (lanes 0..31: tile 0, lanes 32..63: tile 1)
                unsigned lo0 = b16_pack2_as(v[0][0], v[0][1], p.out_f16), lo1 = b16_pack2_as(v[1][0], v[1][1], p.out_f16);
                unsigned hi0 = b16_pack2_as(v[0][2], v[0][3], p.out_f16), hi1 = b16_pack2_as(v[1][2], v[1][3], p.out_f16);
                swap_lane_halves(lo0, lo1);
                swap_lane_halves(hi0, hi1);
                if ((half ? ok[1] : ok[0]) && kblk + blk < CBo) {
                    char* o = static_cast<char*>(p.out) + (((size_t)n * CBo + kblk + blk) * OHW + (half ? pix[1] : pix[0])) * 16;
                    *reinterpret_cast<uint4*>(o) = uint4{lo0, hi0, lo1, hi1};
                }
            } else if (ok[0] && kblk + blk < CBo) {
                uint2 w2;
                w2.x = b16_pack2_as(v[0][0], v[0][1], p.out_f16);
                w2.y = b16_pack2_as(v[0][2], v[0][3], p.out_f16);
                char* o = static_cast<char*>(p.out) + (((size_t)n * CBo + kblk + blk) * OHW + pix[0]) * 16 + half * 8;
                *reinterpret_cast<uint2*>(o) = w2;
            }
        }
}

// The layer's epilogue followed by a 1x1 convolution to at most four channels, on the fp32 values before they would
// have been rounded for storage: a lane holds MT*16 of its pixel's channels (lanes l and l+32 the two halves), so
// the head is MT*16 multiply-adds per output and lane and one exchange between the lane halves.  Lane half h then
// applies the head's activation to outputs h and h+2 and stores them (fp32, planar or pixel-interleaved).
template <int MT, int NT, bool PAIR, bool FOLD, int HP>
__device__ __forceinline__ void b16_head_epilogue(const b16_f32x16 (&acc)[MT][NT], const B16Args& p, const B16Phase& ph,
                                                  const float* s_par, int n, int oh, int j0, int half, int j) {
    static_assert(HP == 2 || HP == kB16MaxHead, "head rows are read in pairs");
    const size_t OHW = (size_t)p.OH * p.OW;
    float sum[NT][HP];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pp = 0; pp < HP; ++pp) sum[nt][pp] = 0.f;
    const float slope0 = s_par[3 * MT * 32];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m0 = (mt * 4 + q) * 8 + half * 4;
            __builtin_amdgcn_sched_barrier(0);          // one channel quad at a time: keeps the parameter reads of later quads out of the registers
            b16_f32x4 hw[HP];
#pragma unroll
            for (int pp = 0; pp < HP; ++pp) hw[pp] = *reinterpret_cast<const b16_f32x4*>(s_par + (5 + pp) * MT * 32 + m0);
            b16_f32x4 P0 = *reinterpret_cast<const b16_f32x4*>(s_par + m0);
            b16_f32x4 P1 = *reinterpret_cast<const b16_f32x4*>(s_par + MT * 32 + m0);
            b16_f32x4 P2, P3, P4;
            if constexpr (!FOLD) {
                P2 = *reinterpret_cast<const b16_f32x4*>(s_par + 2 * MT * 32 + m0);
                P3 = *reinterpret_cast<const b16_f32x4*>(s_par + 3 * MT * 32 + m0);
                P4 = *reinterpret_cast<const b16_f32x4*>(s_par + 4 * MT * 32 + m0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float x;
                    if constexpr (FOLD) {
                        x = fmaf(acc[mt][nt][q * 4 + r], P0[r], P1[r]);
                        x = x >= 0.f ? x : x * slope0;
                    } else {
                        x = acc[mt][nt][q * 4 + r] + P0[r];
                        if (p.act_first) x = (x >= 0.f ? x : x * P3[r]) * P1[r] + P2[r];
                        else {
                            x = x * P1[r] + P2[r];
                            x = x >= 0.f ? x : x * P3[r];
                        }
                        x *= P4[r];
                    }
#pragma unroll
                    for (int pp = 0; pp < HP; ++pp) sum[nt][pp] = fmaf(x, hw[pp][r], sum[nt][pp]);
                }
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int jc = PAIR ? j0 + j : j0 + nt * 32 + j;
        const bool ok = jc < ph.cols[PAIR ? nt : 0];
        const size_t pix = (size_t)oh * p.OW + (size_t)(jc * p.sow + ph.pow_[PAIR ? nt : 0]);
#pragma unroll
        for (int k = 0; k < HP / 2; ++k) {
            // outputs 2k and 2k+1: the lower lanes finish output 2k, the upper lanes output 2k+1 -- each lane sends
            // the partial sum the OTHER half needs and keeps the one it finishes itself (through ds_bpermute)
            const float send = half ? sum[nt][2 * k] : sum[nt][2 * k + 1];
            const float keep = half ? sum[nt][2 * k + 1] : sum[nt][2 * k];
            const float recv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((half ^ 1) * 32 + j) << 2,
                                                                                      __builtin_bit_cast(int, send)));
            const int pp = 2 * k + half;
            if (!ok || pp >= p.head_P) continue;
            float y = keep + recv + (p.head_b ? p.head_b[pp] : 0.f);
            y = b16_head_act(y, p.head_act);
            float* o = static_cast<float*>(p.out);
            if (p.head_il) o[((size_t)n * OHW + pix) * p.head_P + pp] = y;
            else o[((size_t)n * p.head_P + pp) * OHW + pix] = y;
        }
    }
}

// OCC2: built for two workgroups per CU (128 registers, LDS images <= 80 KB, input image <= 32 KB), so one
// workgroup's prologue and epilogue run under the other's matrix loop.
//
// RT: output rows per wave.  With RT = 2 a wave owns rows w and w + 8 of a 16-row workgroup tile: MT x (RT*NT)
// accumulator tiles, so a tap's MT weight operands serve RT*NT pixel operands (MT = 3, NT = 2: 7 LDS reads per 12
// MFMAs instead of 5 per 6 -- 0.58 KB of LDS reads per MFMA against a budget of 1 KB at the matrix peak) and the
// input rows of the taps' halo are shared by twice as many output rows.  One workgroup per CU (192 accumulators).
template <int MT, int NT, bool GEN, bool OCC2, bool PAIR, int RT = 1, bool F16 = false>
__device__ __forceinline__ void conv_b16_body(const B16Args& p) {
    static_assert(!PAIR || NT == 2, "pair mode uses both column tiles");
    static_assert(RT == 1 || (!OCC2 && !GEN && NT == 2), "two rows per wave: the one-workgroup-per-CU two-tile kernels only");
    constexpr int WROWS = kB16Waves * RT;               // phase-grid rows of a workgroup tile
    constexpr int MAXQ = OCC2 ? kB16MaxQ / 2 : kB16MaxQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int nin_buf = p.din + 1;
    char* const s_in = smem;
    char* const s_w = smem + nin_buf * p.in_buf_bytes;
    float* const s_par = reinterpret_cast<float*>(s_w + p.wring * p.w_buf_bytes);     // [5 (+ 4 with a head)][MT*32]

    // ---- this workgroup's phase and tile ----------------------------------------------
    unsigned bid = blockIdx.x;
    int a = 0;
    unsigned start = 0;
#pragma unroll
    for (int q = 0; q < kB16MaxPhase - 1; ++q)
        if (q + 1 < p.nphase && bid >= p.ph[q].wg_end) { a = q + 1; start = p.ph[q].wg_end; }
    bid -= start;
    constexpr int TW = PAIR ? 32 : 32 * NT;            // columns of the phase grid per workgroup
    const int nth = p.ph[a].nth, ntw0 = p.ph[a].ntw[0], ntw1 = p.ph[a].ntw[1];
    const int ntws = PAIR ? ntw0 + ntw1 : ntw0;         // taps per tap row in the weight image
    const int prow = p.ph[a].rows;
    const int njt = (p.ph[a].cols[0] + TW - 1) / TW, nit = (prow + WROWS - 1) / WROWS;
    const int jt = bid % njt; bid /= njt;
    const int it = bid % nit;
    const int n = bid / nit;
    const int i0 = it * WROWS, j0 = jt * TW;

    const int spanh = (nth - 1) * p.tdh;
    const int negh = spanh < 0 ? spanh : 0;
    const int rows_in = (WROWS - 1) * p.sih + (spanh < 0 ? -spanh : spanh) + 1;
    // columns of the input image: the union of what the column tiles read
    int wbase, wtop;
    {
        const int span0 = (ntw0 - 1) * p.tdw, span1 = (ntw1 - 1) * p.tdw;
        const int lo0 = p.ph[a].offw[0] + (span0 < 0 ? span0 : 0), lo1 = p.ph[a].offw[1] + (span1 < 0 ? span1 : 0);
        const int hi0 = (TW - 1) * p.siw + p.ph[a].offw[0] + (span0 > 0 ? span0 : 0);
        const int hi1 = (TW - 1) * p.siw + p.ph[a].offw[1] + (span1 > 0 ? span1 : 0);
        wbase = PAIR ? (lo0 < lo1 ? lo0 : lo1) : lo0;
        wtop = PAIR ? (hi0 > hi1 ? hi0 : hi1) : hi0;
    }
    const int cols_in = wtop - wbase + 1;
    const int plane = rows_in * cols_in;
    const int slots = 2 * plane;
    const int n_in_instr = (slots + 63) >> 6;
    const int ihb = i0 * p.sih + p.ph[a].offh + negh;
    const int iwb = j0 * p.siw + wbase;
    const int HW = p.H * p.W;

    // ---- per-lane byte offsets of this wave's input DMA instructions (same for every group) ----
    int voff[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int e = (wave + kB16Waves * k) * 64 + lane;
        int off = (int)0x80000000;
        if (e < slots) {
            const int hf = e >= plane ? 1 : 0;
            const int rem = e - hf * plane;
            const int r = rem / cols_in, c = rem - r * cols_in;
            const int ih = ihb + r, iw = iwb + c;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) off = (hf * HW + ih * p.W + iw) * 16;
        }
        voff[k] = off;
    }

    const int G = p.G0 + p.G1;
    const int R = p.allrows ? nth : 1;               // tap rows per step
    const int nchunk = nth / R;
    const int step_bytes = R * ntws * MT * 1024;
    const int n_w_instr = R * ntws * MT;
    const int nsub = p.hl * nchunk;                  // steps per channel group: weight half (outer) x tap-row chunk
    const int S = G * nsub;
    const int kb = (int)blockIdx.y * 128;            // this workgroup's chunk of a layer with more than 128 outputs
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(p.w + (size_t)blockIdx.y * p.w_chunk_bytes + p.ph[a].w_off), 0, (int)p.ph[a].w_bytes, 0x00020000);

    auto issue_input = [&](int g) -> int {            // group g -> its ring slot; returns instructions issued
        if (g >= G) return 0;
        const bool second = g >= p.G0;
        const int gl = second ? g - p.G0 : g;
        const int CB = 2 * (second ? p.G1 : p.G0);
        const char* base = (second ? p.in1 : p.in0) + (size_t)n * CB * HW * 16;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, CB * HW * 16, 0x00020000);
        const int soff = 2 * gl * HW * 16;
        char* dst = s_in + (g % nin_buf) * p.in_buf_bytes;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            const int q = wave + kB16Waves * k;
            if (q < n_in_instr) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (b16_lds_ptr)(dst + q * 1024), 16, voff[k], soff, 0, 0);
                ++cnt;
            }
        }
        return cnt;
    };
    auto issue_weights = [&](int s) -> int {          // step s -> ring slot s % wring
        if (s >= S) return 0;
        char* dst = s_w + (s % p.wring) * p.w_buf_bytes;
        const int soff0 = s * step_bytes;
        int cnt = 0;
        for (int q = wave; q < n_w_instr; q += kB16Waves) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (b16_lds_ptr)(dst + q * 1024), 16, lane * 16, soff0 + q * 1024, 0, 0);
            ++cnt;
        }
        return cnt;
    };

    // ---- prologue: epilogue parameters and the first steps' operands --------------------------
    // Can this workgroup's epilogue take the folded form (see b16_epilogue)?  One slope for all channels, affine
    // before the activation, and no negative gate value for this sample.
    float gv = 1.f;
    if (tid < p.K && p.gate) gv = p.gate[(size_t)n * p.Kfull + kb + tid];
    int* const s_flag = reinterpret_cast<int*>(s_par + (5 + (p.head_P ? kB16MaxHead : 0)) * MT * 32);
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    if (gv < 0.f) *s_flag = 1;                          // every writer stores the same value
    __syncthreads();
    const bool neg_gate = __builtin_amdgcn_readfirstlane(*s_flag) != 0;
    const bool fold = !GEN && !p.act_first && !p.act_pc && !neg_gate;
    if (tid < MT * 32) {
        float par[5] = {0.f, 0.f, 0.f, 0.f, 0.f};       // channels past K produce exact zeros
        if (tid < p.K) {
            float slope = p.act == UNETMI_ACT_NONE ? 1.f : 0.f;                  // slope of the select: none = 1, ReLU = 0
            if (p.act == UNETMI_ACT_LEAKY) slope = p.act_param[p.act_pc ? kb + tid : 0];
            par[3] = slope;
            if (fold) {
                par[0] = p.scale[kb + tid] * gv;
                par[1] = (p.bias[kb + tid] * p.scale[kb + tid] + p.shift[kb + tid]) * gv;
            } else {
                par[0] = p.bias[kb + tid]; par[1] = p.scale[kb + tid]; par[2] = p.shift[kb + tid]; par[4] = gv;
            }
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) s_par[q * MT * 32 + tid] = par[q];
    }
    if (p.head_P)
        for (int i = tid; i < kB16MaxHead * MT * 32; i += 512) {
            const int pp = i / (MT * 32), c = i - pp * (MT * 32);
            s_par[5 * MT * 32 + i] = (pp < p.head_P && c < p.K) ? p.head_w[pp * p.K + c] : 0.f;
        }
    for (int g = 0; g < p.din; ++g) issue_input(g);
    issue_weights(0);
    if (p.wring == 3) issue_weights(1);
    __syncthreads();                                   // full drain once

    b16_f32x16 acc[RT][MT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rt][mt][nt][r] = 0.f;

    int colb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        colb[nt] = ((PAIR ? j : nt * 32 + j) * p.siw + p.ph[a].offw[PAIR ? nt : 0] - wbase) * 16;
    // per-lane LDS byte addresses: this lane's slot of a weight operand, and its pixel of an input image's first row
    const int la = b16_lds_addr(s_w) + half * 512 + j * 16;
    int lb[RT][NT];                                    // row tile rt: 8 phase-grid rows = 8 * sih input rows further down
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            lb[rt][nt] = b16_lds_addr(s_in) + half * plane * 16 + colb[nt] + rt * kB16Waves * p.sih * cols_in * 16;

    int s = 0;
    for (int g = 0; g < G; ++g) {
        for (int sub = 0; sub < nsub; ++sub, ++s) {
            const int tc = sub % nchunk;
            int issued = (B16_ABLATE & 2) ? 0 : issue_weights(s + p.wring - 1);
            if (sub == 0 && !(B16_ABLATE & 2)) issued += issue_input(g + p.din);

            // The operand reads of a step are software-pipelined inside each wave: the sequence of weight operands
            // (tap, mt) alternates between two register sets, each read issued one MFMA group before its use, and
            // the pixel operands of the next tap are requested as soon as the last MFMA group of the current tap has
            // issued -- a wave that waited for every ds_read_b128 in front of the MFMAs that use it spent more time
            // on the ~15 exposed LDS round trips of a step than on the step's MFMAs.
            // An LDS address is (per-lane base) + (wave-uniform offset): the uniform offsets are kept opaque to the
            // optimizer (b16_keep_scalar), which otherwise strength-reduces the tap walk into a dozen live per-lane
            // pointers and spills the DMA offsets to make room for them.
            int uw = b16_keep_scalar((s % p.wring) * p.w_buf_bytes);                        // weights of the next A read
            const int rowl0 = wave * p.sih + tc * R * p.tdh - negh;
            const int ub0 = (g % nin_buf) * p.in_buf_bytes + rowl0 * cols_in * 16;         // tap (row 0, column 0)
            const int tdw16 = p.tdw * 16, rowstep = p.tdh * cols_in * 16;
            if constexpr (PAIR) {
                // a segment = the taps of one column phase in one tap row; nt is compile-time (it names accumulators)
                for (int tr = 0; tr < ((B16_ABLATE & 1) ? 0 : R); ++tr) {
                    static_for_b16<0, 2>([&](auto nt_tag) {
                        constexpr int nt = decltype(nt_tag)::value;
                        const int cnt = nt ? ntw1 : ntw0;
                        int ub = b16_keep_scalar(ub0 + tr * rowstep);
                        bf16x8 A[2], B[RT];
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) B[rt] = b16_lds_ld(lb[rt][nt], ub, 0);
                        A[0] = b16_lds_ld(la, uw, 0);
                        auto tap = [&](auto par_tag, bool more) {
                            constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const int cur = (PAR + mt) & 1;
                                if (mt + 1 < MT) A[cur ^ 1] = b16_lds_ld(la, uw, (mt + 1) * 1024);
                                else if (more) A[cur ^ 1] = b16_lds_ld(la, uw, MT * 1024);
                                __builtin_amdgcn_sched_barrier(0);      // the read goes out BEFORE the MFMAs it hides under
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt)
                                    acc[rt][mt][nt] = b16_mfma<F16>(A[cur], B[rt], acc[rt][mt][nt]);
                            }
                            uw = b16_keep_scalar(uw + MT * 1024);
                            ub = b16_keep_scalar(ub + tdw16);
                            if (more) {
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt) B[rt] = b16_lds_ld(lb[rt][nt], ub, 0);
                            }
                        };
                        int u = 0;
                        for (; u + 1 < cnt; u += 2) {
                            tap(std::integral_constant<int, 0>{}, true);
                            tap(std::integral_constant<int, (MT & 1)>{}, u + 2 < cnt);
                        }
                        if (u < cnt) tap(std::integral_constant<int, 0>{}, false);
                    });
                }
            } else {
                const int T = (B16_ABLATE & 1) ? 0 : R * ntw0;      // taps of this step, row-major
                if (T > 0) {
                    int ub = b16_keep_scalar(ub0), twn = 0;         // pixel operands and tap column of the NEXT B read
                    auto next_ub = [&]() {
                        int nb = ub + tdw16;
                        if (++twn == ntw0) { twn = 0; nb += rowstep - ntw0 * tdw16; }
                        ub = b16_keep_scalar(nb);
                    };
                    bf16x8 A[2], B[RT][NT];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) B[rt][nt] = b16_lds_ld(lb[rt][nt], ub, 0);
                    A[0] = b16_lds_ld(la, uw, 0);
                    next_ub();
                    auto tap = [&](auto par_tag, bool more) {
                        constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const int cur = (PAR + mt) & 1;
                            if (mt + 1 < MT) A[cur ^ 1] = b16_lds_ld(la, uw, (mt + 1) * 1024);
                            else if (more) A[cur ^ 1] = b16_lds_ld(la, uw, MT * 1024);
                            __builtin_amdgcn_sched_barrier(0);          // the read goes out BEFORE the MFMAs it hides under
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[rt][mt][nt] = b16_mfma<F16>(A[cur], B[rt][nt], acc[rt][mt][nt]);
                        }
                        if (more) {
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) B[rt][nt] = b16_lds_ld(lb[rt][nt], ub, 0);
                        }
                        uw = b16_keep_scalar(uw + MT * 1024);
                        next_ub();
                    };
                    int u = 0;
                    for (; u + 1 < T; u += 2) {
                        tap(std::integral_constant<int, 0>{}, true);
                        tap(std::integral_constant<int, (MT & 1)>{}, u + 2 < T);
                    }
                    if (u < T) tap(std::integral_constant<int, 0>{}, false);
                }
            }
            // everything issued before this step has to have landed; this step's own loads may fly on
            if (!(B16_ABLATE & 8)) {
                b16_wait_vm(p.wring == 3 ? __builtin_amdgcn_readfirstlane(issued) : 0);
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------
    if ((B16_ABLATE & 4) && p.N >= 0) return;           // timing experiment: no epilogue
    static_for_b16<0, RT>([&](auto rt_tag) {
        constexpr int rt = decltype(rt_tag)::value;     // compile-time: names accumulator registers
        const int i = i0 + rt * kB16Waves + wave;
        if (i >= prow) return;                          // wave-uniform
        const int oh = i * p.soh + p.ph[a].poh;
        if constexpr (!GEN && NT == 1 && !PAIR) {
            // the fused head exists in the one-column-tile kernels only (the plan gives a layer with a head one of
            // them): four more epilogue bodies in every two-tile kernel made the compiler spill hundreds of registers
            if (p.head_P) {
                if (p.head_P <= 2) {
                    if (fold) b16_head_epilogue<MT, NT, PAIR, true, 2>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j);
                    else b16_head_epilogue<MT, NT, PAIR, false, 2>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j);
                } else {
                    if (fold) b16_head_epilogue<MT, NT, PAIR, true, kB16MaxHead>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j);
                    else b16_head_epilogue<MT, NT, PAIR, false, kB16MaxHead>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j);
                }
                return;
            }
        }
        if constexpr (!GEN) {
            if (fold) {
                b16_epilogue<MT, NT, false, PAIR, true>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j, kb);
                return;
            }
        }
        b16_epilogue<MT, NT, GEN, PAIR, false>(acc[rt], p, p.ph[a], s_par, n, oh, j0, half, j, kb);
    });
}

// F16: the operands (activations and weight image) are fp16 instead of bfloat16
template <int MT, int NT, bool GEN, bool F16>
__global__ __launch_bounds__(512, 2) void conv_b16_kernel(const B16Args p) { conv_b16_body<MT, NT, GEN, false, false, 1, F16>(p); }

template <int MT, int NT, bool F16>
__global__ __launch_bounds__(512, 4) void conv_b16_kernel_occ2(const B16Args p) { conv_b16_body<MT, NT, false, true, false, 1, F16>(p); }

template <int MT, bool F16>
__global__ __launch_bounds__(512, 2) void conv_b16_kernel_pair(const B16Args p) { conv_b16_body<MT, 2, false, false, true, 1, F16>(p); }

template <int MT, bool F16>
__global__ __launch_bounds__(512, 4) void conv_b16_kernel_pair_occ2(const B16Args p) { conv_b16_body<MT, 2, false, true, true, 1, F16>(p); }

// two output rows per wave (16-row workgroup tiles), two column tiles / the two column phases
template <int MT, bool F16>
__global__ __launch_bounds__(512, 2) void conv_b16_kernel_rt2(const B16Args p) { conv_b16_body<MT, 2, false, false, false, 2, F16>(p); }

template <int MT, bool F16>
__global__ __launch_bounds__(512, 2) void conv_b16_kernel_pair_rt2(const B16Args p) { conv_b16_body<MT, 2, false, false, true, 2, F16>(p); }

// ---- layout conversion ------------------------------------------------------------------------
// NCHW (fp32 or bfloat16) -> blocked; a thread owns one pixel of one 8-channel block.
// A thread owns FOUR consecutive pixels of one 8-channel block: per channel one 16-byte (fp32) or 8-byte (bfloat16)
// load -- element alignment is enough for them -- and four 16-byte records out (64 contiguous bytes per thread).
template <bool SRC_BF16>
__global__ __launch_bounds__(256) void to_blocked_kernel(const void* src, const void* src2, char* dst, int C, int CB, int HW,
                                                         int dst_f16) {
    const int pix = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int blk = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const int np = min(4, HW - pix);
    float f[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = blk * 8 + k;
#pragma unroll
        for (int q = 0; q < 4; ++q) f[k][q] = 0.f;
        if (c >= C) continue;
        const size_t base = ((size_t)n * C + c) * HW + pix;
        if (np == 4) {
            if constexpr (SRC_BF16) {
                const uint2 w = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(src) + base);
                f[k][0] = __builtin_bit_cast(float, w.x << 16); f[k][1] = __builtin_bit_cast(float, w.x & 0xffff0000u);
                f[k][2] = __builtin_bit_cast(float, w.y << 16); f[k][3] = __builtin_bit_cast(float, w.y & 0xffff0000u);
                if (src2) {
                    const uint2 w2 = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(src2) + base);
                    f[k][0] += __builtin_bit_cast(float, w2.x << 16); f[k][1] += __builtin_bit_cast(float, w2.x & 0xffff0000u);
                    f[k][2] += __builtin_bit_cast(float, w2.y << 16); f[k][3] += __builtin_bit_cast(float, w2.y & 0xffff0000u);
                }
            } else {
                const b16_f32x4 w = *reinterpret_cast<const b16_f32x4_u*>(static_cast<const float*>(src) + base);
#pragma unroll
                for (int q = 0; q < 4; ++q) f[k][q] = w[q];
                if (src2) {
                    const b16_f32x4 w2 = *reinterpret_cast<const b16_f32x4_u*>(static_cast<const float*>(src2) + base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[k][q] += w2[q];          // summed in fp32, rounded once
                }
            }
        } else {
            for (int q = 0; q < np; ++q) {
                f[k][q] = ld_act<SRC_BF16>(src, base + q);
                if (src2) f[k][q] += ld_act<SRC_BF16>(src2, base + q);
            }
        }
    }
    char* o = dst + (((size_t)n * CB + blk) * HW + pix) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= np) break;
        uint4 r;
        r.x = b16_pack2_as(f[0][q], f[1][q], dst_f16); r.y = b16_pack2_as(f[2][q], f[3][q], dst_f16);
        r.z = b16_pack2_as(f[4][q], f[5][q], dst_f16); r.w = b16_pack2_as(f[6][q], f[7][q], dst_f16);
        *reinterpret_cast<uint4*>(o + q * 16) = r;
    }
}

template <bool DST_BF16>
__global__ __launch_bounds__(256) void from_blocked_kernel(const char* src, void* dst, int C, int CB, int HW, int src_f16) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int blk = blockIdx.y, n = blockIdx.z;
    if (pix >= HW) return;
    const uint4 o = *reinterpret_cast<const uint4*>(src + (((size_t)n * CB + blk) * HW + pix) * 16);
    const unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = blk * 8 + k;
        if (c >= C) continue;
        const unsigned short b = (unsigned short)(k & 1 ? w[k >> 1] >> 16 : w[k >> 1] & 0xffffu);
        const float v = src_f16 ? (float)__builtin_bit_cast(_Float16, b) : bf16_bits_to_f32(b);
        st_act<DST_BF16>(dst, ((size_t)n * C + c) * HW + pix, v);
    }
}

// ---- host: plan, packer, launch ---------------------------------------------------------------
struct B16Plan {
    B16Args args;
    int MT, NT, RT;
    int kchunks;             // 128-channel output chunks (grid dimension y)
    bool occ2;
    unsigned grid;
    int kw0[kB16MaxPhase][2];
    size_t lds_bytes, packed_bytes;
    int kh0[kB16MaxPhase], khs, kws;                         // kernel index of tap t: k0 + t * ks
};

static unsigned short host_bf16(float v) {
    unsigned u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float host_bf16_to_f32(unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float v;
    std::memcpy(&v, &u, 4);
    return v;
}

// fp32 -> fp16 bits for the weight image, nearest-even; a weight beyond the largest finite value saturates
static unsigned short host_f16(float v) {
    unsigned u;
    std::memcpy(&u, &v, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const unsigned mag = u & 0x7fffffffu;
    if (mag > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);            // NaN
    if (mag >= 0x477ff000u) return (unsigned short)(sign | 0x7bffu);           // >= 65520 rounds past 65504: saturate
    if (mag < 0x33000001u) return (unsigned short)sign;                        // below half the smallest subnormal
    int e = (int)(mag >> 23) - 127;
    unsigned m = (mag & 0x7fffffu) | 0x800000u;                                // 24-bit significand
    int shift = e < -14 ? 13 + (-14 - e) : 13;                                  // bits dropped (subnormals drop more)
    unsigned half = m >> shift;
    const unsigned rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    // half holds the significand with its hidden bit (normal) or without (subnormal); a carry moves into the exponent
    const unsigned bits = e < -14 ? half : ((unsigned)(e + 15 - 1) << 10) + half;
    return (unsigned short)(sign | bits);
}
static float host_f16_to_f32(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const unsigned m = h & 0x3ffu;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m | 0x400u), e - 25);
    unsigned u;
    std::memcpy(&u, &v, 4);
    u |= sign;
    std::memcpy(&v, &u, 4);
    return v;
}

static bool make_b16_plan(const UnetmiConvDesc& d, B16Plan* pl, std::string* err, int head_P = 0) {
    auto bad = [&](const char* m) { *err = m; return false; };
    if (d.N < 0 || d.C0 < 1 || d.C1 < 0 || d.H < 1 || d.W < 1 || d.K < 1 || d.kh < 1 || d.kw < 1 || d.sh < 1 || d.sw < 1 ||
        d.dh < 1 || d.dw < 1 || d.ph < 0 || d.pw < 0)
        return bad("conv2d_b16: bad sizes");
    if (d.in_mode != UNETMI_IN_SINGLE && d.in_mode != UNETMI_IN_CONCAT) return bad("conv2d_b16: in_mode must be single or concat");
    if (d.in_mode == UNETMI_IN_SINGLE && d.C1 != 0) return bad("conv2d_b16: C1 must be 0 for a single input");
    if (d.has_residual) return bad("conv2d_b16: no residual input");
    const int hl = d.reserved[1] == 0 ? 1 : d.reserved[1];
    if (hl != 1 && hl != 2) return bad("conv2d_b16: reserved[1] (weight halves) must be 0, 1 or 2");
    int OH, OW;
    if (d.transposed) {
        if ((d.sh > 1 && d.dh != 1) || (d.sw > 1 && d.dw != 1)) return bad("conv2d_b16: strided transposed conv needs dilation 1");
        OH = (d.H - 1) * d.sh - 2 * d.ph + d.dh * (d.kh - 1) + d.oph + 1;
        OW = (d.W - 1) * d.sw - 2 * d.pw + d.dw * (d.kw - 1) + d.opw + 1;
    } else {
        OH = (d.H + 2 * d.ph - d.dh * (d.kh - 1) - 1) / d.sh + 1;
        OW = (d.W + 2 * d.pw - d.dw * (d.kw - 1) - 1) / d.sw + 1;
    }
    if (OH != d.OH || OW != d.OW || OH < 1 || OW < 1) return bad("conv2d_b16: OH/OW do not match the geometry");
    B16Args& a = pl->args;
    std::memset(&a, 0, sizeof(a));
    // more than 128 output channels: chunks of 128 as a second grid dimension (every chunk the same kernel and
    // weight-image shape, hence whole multiples only); no fused head there
    if (d.K > 128 && (d.K % 128 != 0 || head_P)) return bad("conv2d_b16: more than 128 output channels come in multiples of 128 (and without a fused head)");
    pl->kchunks = d.K > 128 ? d.K / 128 : 1;
    if (pl->kchunks > 65535) return bad("conv2d_b16: too many output channels");
    const int Kc = d.K > 128 ? 128 : d.K;
    a.N = d.N; a.H = d.H; a.W = d.W; a.OH = OH; a.OW = OW; a.K = Kc; a.Kfull = d.K;
    a.G0 = (d.C0 + 15) / 16; a.G1 = (d.C1 + 15) / 16;
    a.hl = hl;
    a.act = d.act; a.act_first = d.act_first; a.act_pc = d.act_param_per_channel; a.out_nchw_f32 = (d.reserved[2] & UNETMI_B16_OUT_NCHW_F32) != 0;
    a.out_f16 = (d.reserved[2] & UNETMI_B16_OUT_F16) != 0;
    if (d.reserved[2] & ~(UNETMI_B16_OUT_NCHW_F32 | UNETMI_B16_IN_F16 | UNETMI_B16_OUT_F16))
        return bad("conv2d_b16: unknown bits in reserved[2]");
    pl->MT = (Kc + 31) / 32;
    if ((size_t)2 * std::max(a.G0, a.G1) * d.H * d.W * 16 >= (size_t)1 << 31) return bad("conv2d_b16: sample larger than 2 GB");
    int np = 0;
    // a time-stride-2 transposed layer runs its two column phases as the two column tiles of a workgroup
    // (geometry only: the weight image depends on it, and the packer does not know the activation)
    a.pair = (d.transposed && d.sw == 2 && pl->MT <= 3) ? 1 : 0;
    if (!d.transposed) {
        a.sih = d.sh; a.siw = d.sw; a.tdh = d.dh; a.tdw = d.dw; a.soh = 1; a.sow = 1;
        B16Phase& p = a.ph[0];
        p.nth = d.kh; p.offh = -d.ph; p.poh = 0; p.rows = OH;
        for (int c = 0; c < 2; ++c) { p.ntw[c] = d.kw; p.offw[c] = -d.pw; p.pow_[c] = 0; p.cols[c] = OW; pl->kw0[0][c] = 0; }
        pl->kh0[0] = 0; pl->khs = 1; pl->kws = 1;
        np = 1;
    } else {
        if (d.sh * d.sw > kB16MaxPhase) return bad("conv2d_b16: more than 4 output phases");
        a.sih = 1; a.siw = 1; a.soh = d.sh; a.sow = d.sw;
        // stride 1 on an axis: one phase holding every tap, walked backwards with the dilation as step
        a.tdh = d.sh == 1 ? -d.dh : -1; a.tdw = d.sw == 1 ? -d.dw : -1;
        pl->khs = d.sh; pl->kws = d.sw;
        for (int ah = 0; ah < d.sh; ++ah)
            for (int aw = 0; aw < (a.pair ? 1 : d.sw); ++aw) {
                B16Phase& p = a.ph[np];
                const int kh0 = (ah + d.ph) % d.sh;
                p.nth = kh0 < d.kh ? (d.kh - kh0 + d.sh - 1) / d.sh : 0;
                // oh = ih*s - p + k*d  ->  ih = (oh + p - k*d) / s ; with oh = i*s + a and k = k0 + t*s (d = 1 when s > 1)
                p.offh = d.sh == 1 ? d.ph : (ah + d.ph - kh0) / d.sh;
                p.poh = ah;
                p.rows = OH > ah ? (OH - ah + d.sh - 1) / d.sh : 0;
                pl->kh0[np] = kh0;
                for (int c = 0; c < 2; ++c) {
                    const int awc = a.pair ? c : aw;
                    const int kw0 = (awc + d.pw) % d.sw;
                    p.ntw[c] = kw0 < d.kw ? (d.kw - kw0 + d.sw - 1) / d.sw : 0;
                    p.offw[c] = d.sw == 1 ? d.pw : (awc + d.pw - kw0) / d.sw;
                    p.pow_[c] = awc;
                    p.cols[c] = OW > awc ? (OW - awc + d.sw - 1) / d.sw : 0;
                    pl->kw0[np][c] = kw0;
                    if (p.ntw[c] < 1) return bad("conv2d_b16: an output phase without taps (kernel smaller than stride)");
                }
                if (p.nth < 1) return bad("conv2d_b16: an output phase without taps (kernel smaller than stride)");
                ++np;
            }
    }
    a.nphase = np;
    if (head_P && a.pair) return bad("conv2d_b16_head: not built for time-stride-2 transposed layers");
    // Tile shape and pipeline depth.  Preference: two workgroups per CU (weight ring of 2, LDS <= 80 KB each,
    // one workgroup's prologue / epilogue hidden under the other's matrix loop) with two column tiles, then with
    // one; else one workgroup per CU with a ring of 3.  A step takes every tap row of a channel group when
    // those weights are at most 24 KB (kernels one column wide would otherwise hit a barrier every few MFMAs).
    int max_nth = 0, max_ntws = 0;                 // max_ntws: taps per tap row in the weight image
    bool one_row_phase = false;
    for (int q = 0; q < np; ++q) {
        max_nth = std::max(max_nth, a.ph[q].nth);
        max_ntws = std::max(max_ntws, a.pair ? a.ph[q].ntw[0] + a.ph[q].ntw[1] : a.ph[q].ntw[0]);
        if (a.ph[q].nth == 1) one_row_phase = true;
    }
    // Few workgroups (the deep layers): nothing else runs on a CU to cover a step's DMA latency, so take fewer,
    // larger steps -- a whole channel group at a time while its weights stay within 48 KB.
    unsigned long long wg64 = 0;
    for (int q = 0; q < np; ++q)
        wg64 += (unsigned long long)d.N * ((a.ph[q].rows + kB16Waves - 1) / kB16Waves) * ((a.ph[q].cols[0] + 63) / 64);
    const bool few_wg = wg64 <= 512;
    const bool allrows_ok = max_nth > 1 && max_nth * max_ntws * pl->MT <= (few_wg ? 48 : 24);
    auto in_bytes = [&](int nt, int rt = 1) {      // largest input image of a channel group over the phases
        const int tw_cols = a.pair ? 32 : 32 * nt;
        int cols_in = 1;
        for (int q = 0; q < np; ++q) {
            int lo = 1 << 30, hi = -(1 << 30);
            for (int c = 0; c < (a.pair ? 2 : 1); ++c) {
                const int span = (a.ph[q].ntw[c] - 1) * a.tdw;
                lo = std::min(lo, a.ph[q].offw[c] + std::min(span, 0));
                hi = std::max(hi, (tw_cols - 1) * a.siw + a.ph[q].offw[c] + std::max(span, 0));
            }
            cols_in = std::max(cols_in, hi - lo + 1);
        }
        const int rows_in = (kB16Waves * rt - 1) * a.sih + (max_nth - 1) * std::abs(a.tdh) + 1;
        return (size_t)(((2 * rows_in * cols_in + 63) / 64) * 1024);
    };
    const size_t par_bytes = (size_t)(5 + (head_P ? kB16MaxHead : 0)) * pl->MT * 32 * 4 + 16;   // + the prologue's flag word
    const bool general_act = d.act != UNETMI_ACT_NONE && d.act != UNETMI_ACT_LEAKY && d.act != UNETMI_ACT_RELU;
    bool found = false;
    pl->RT = 1;
    for (int cc = allrows_ok ? 0 : 6; cc < 12 && !found; ++cc) {      // first with whole-group steps, then without
        const int c = cc % 6;                       // 4, 5: one workgroup per CU with rings of 2 (large steps)
        a.allrows = cc < 6 ? 1 : 0;
        a.w_buf_bytes = (a.allrows ? max_nth : 1) * max_ntws * pl->MT * 1024;
        const bool single_step_group = (a.allrows || one_row_phase) && hl == 1;
        const int nt = (c & 1) ? 1 : 2;
        const bool occ2 = c < 2;
        const bool shallow = c >= 4;
        if (nt == 2 && pl->MT > 3) continue;
        if (a.pair && nt != 2) continue;
        if (head_P && nt != 1) continue;            // the fused head is built into the one-column-tile kernels
        if (occ2 && general_act) continue;
        const size_t ib = in_bytes(nt);
        const int din = (occ2 || shallow) ? 1 : (single_step_group ? 2 : 1);
        const int wring = (occ2 || shallow) ? 2 : 3;   // (a ring of 3 with two workgroups per CU was measured: no gain)
        const size_t lds = (size_t)(din + 1) * ib + (size_t)wring * a.w_buf_bytes + par_bytes;
        const bool fits = occ2 ? (lds <= 81856 && ib <= (size_t)kB16MaxQ / 2 * kB16Waves * 1024)
                               : (lds <= 160 * 1024 && ib <= (size_t)kB16MaxQ * kB16Waves * 1024);
        if (!fits) continue;
        pl->NT = nt; pl->occ2 = occ2; a.din = din; a.wring = wring; a.in_buf_bytes = (int)ib; pl->lds_bytes = lds;
        found = true;
    }
    // Two output rows per wave (16-row workgroup tiles, one workgroup per CU: with MT = 3, 7 LDS operand reads per 12
    // MFMAs instead of 5 per 6 or 4 per 3, and a shorter halo).  Measured per layer at batch 32 (tools/b16_probe.py with
    // UNETMI_B16_RT = 1 / 2): it pays where the regular choice is a one-column-tile kernel (a (7,1) layer over 135
    // channels 982 -> 849 us) and, slightly, on time-stride-2 transposed layers with three channel tiles (1429 ->
    // 1391 us); two-column-tile layers with two resident workgroups lose 6 % to it, so they keep the regular plan.
    // UNETMI_B16_RT = 1 / 2 (developer A/B switch, read once): never / wherever the images fit.
    static const int rt_mode = [] { const char* e = std::getenv("UNETMI_B16_RT"); return e ? std::atoi(e) : 0; }();
    if (rt_mode != 1 && !head_P && !general_act && pl->MT <= 3 &&
        (rt_mode == 2 || (pl->MT == 3 && (a.pair || (found && pl->NT == 1))))) {
        int min_rows = 1 << 30;
        unsigned long long wg_rt = 0;
        for (int q = 0; q < np; ++q) {
            min_rows = std::min(min_rows, a.ph[q].rows);
            wg_rt += (unsigned long long)d.N * ((a.ph[q].rows + 2 * kB16Waves - 1) / (2 * kB16Waves)) *
                     ((a.ph[q].cols[0] + (a.pair ? 31 : 63)) / (a.pair ? 32 : 64));
        }
        const size_t ib = in_bytes(2, 2);
        bool taken = false;
        if ((rt_mode == 2 || (min_rows >= 2 * kB16Waves && wg_rt >= 512)) && ib <= (size_t)kB16MaxQ * kB16Waves * 1024) {
            for (int ar = allrows_ok ? 1 : 0; ar >= 0 && !taken; --ar) {
                const int wbuf = (ar ? max_nth : 1) * max_ntws * pl->MT * 1024;
                // A channel group that is ONE step needs its input two groups ahead (a step's counted wait lets the
                // step's own loads fly on, and the next group's input would be one of them): three input images and
                // the ring of 3, else two images, a ring of 2 and a full wait per step.
                const bool single_step_group = (ar || one_row_phase) && hl == 1;
                for (int deep = 1; deep >= 0 && !taken; --deep) {
                    const int din = deep ? (single_step_group ? 2 : 1) : 1, wring = deep ? 3 : 2;
                    const size_t lds = (size_t)(din + 1) * ib + (size_t)wring * wbuf + par_bytes;
                    if (lds > 160 * 1024) continue;
                    a.allrows = ar; a.w_buf_bytes = wbuf;
                    pl->NT = 2; pl->RT = 2; pl->occ2 = false; a.din = din; a.wring = wring; a.in_buf_bytes = (int)ib;
                    pl->lds_bytes = lds;
                    taken = found = true;
                }
            }
        }
    }
    if (!found) return bad("conv2d_b16: LDS images exceed 160 KB");
    // weight image and grid
    const int G = a.G0 + a.G1;
    size_t woff = 0;
    unsigned long long wg = 0;
    for (int q = 0; q < np; ++q) {
        B16Phase& p = a.ph[q];
        p.w_off = (unsigned)woff;
        const int ntws = a.pair ? p.ntw[0] + p.ntw[1] : p.ntw[0];
        p.w_bytes = (unsigned)((size_t)G * p.nth * ntws * hl * pl->MT * 1024);
        woff += p.w_bytes;
        const int tw_cols = a.pair ? 32 : 32 * pl->NT;
        const int wrows = kB16Waves * pl->RT;
        const unsigned long long njt = (p.cols[0] + tw_cols - 1) / tw_cols, nit = (p.rows + wrows - 1) / wrows;
        wg += njt * nit * (unsigned long long)d.N;
        p.wg_end = (unsigned)wg;
    }
    if (wg >= 0x7fffffffull) return bad("conv2d_b16: too many workgroups");
    if (woff >= ((size_t)1 << 31)) return bad("conv2d_b16: weight image larger than 2 GB");
    pl->grid = (unsigned)wg;
    a.w_chunk_bytes = (unsigned)woff;
    pl->packed_bytes = woff * (size_t)pl->kchunks;
    if (pl->packed_bytes >= ((size_t)1 << 32)) return bad("conv2d_b16: weight image larger than 4 GB");
    return true;
}

template <typename Kernel>
static hipError_t launch_b16(Kernel kernel, PerDeviceLatch* raised, const B16Plan& pl, hipStream_t stream) {
    const hipError_t e = raised->once([kernel] {       // dynamic LDS beyond 64 KB: once per kernel and device
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(pl.grid, pl.kchunks), dim3(512), pl.lds_bytes, stream, pl.args);
    return hipGetLastError();
}

}  // namespace unetmi

using namespace unetmi;

extern "C" {

size_t unetmi_conv2d_b16_packed_weight_bytes(const UnetmiConvDesc* d) {
    if (!d) { api_fail(UNETMI_E_BAD_ARG, "null descriptor"); return 0; }
    B16Plan pl;
    std::string err;
    if (!make_b16_plan(*d, &pl, &err)) { api_fail(UNETMI_E_BAD_DESC, err); return 0; }
    return pl.packed_bytes;
}

int unetmi_conv2d_b16_kernel_name(const UnetmiConvDesc* d, int head_channels, char* buf, size_t n) {
    if (!d || !buf || n == 0) return api_fail(UNETMI_E_BAD_ARG, "conv2d_b16_kernel_name: null argument");
    B16Plan pl;
    std::string err;
    if (!make_b16_plan(*d, &pl, &err, head_channels)) return api_fail(UNETMI_E_BAD_DESC, err);
    const bool gen = d->act != UNETMI_ACT_NONE && d->act != UNETMI_ACT_LEAKY && d->act != UNETMI_ACT_RELU;
    std::string name;
    const std::string mt = std::to_string(pl.MT), nt = std::to_string(pl.NT);
    const std::string f16 = (d->reserved[2] & UNETMI_B16_IN_F16) ? ", true>" : ", false>";     // operands fp16 / bfloat16
    if (pl.RT == 2) name = std::string(pl.args.pair ? "conv_b16_kernel_pair_rt2<" : "conv_b16_kernel_rt2<") + mt + f16;
    else if (pl.args.pair) name = std::string(pl.occ2 ? "conv_b16_kernel_pair_occ2<" : "conv_b16_kernel_pair<") + mt + f16;
    else if (pl.occ2) name = "conv_b16_kernel_occ2<" + mt + ", " + nt + f16;
    else name = "conv_b16_kernel<" + mt + ", " + nt + ", " + (gen ? "true" : "false") + f16;
    std::snprintf(buf, n, "%s", name.c_str());
    return UNETMI_OK;
}

int unetmi_conv2d_b16_pack_weights(const UnetmiConvDesc* d, const float* w, void* packed) {
    if (!d || !w || !packed) return api_fail(UNETMI_E_BAD_ARG, "null argument to conv2d_b16_pack_weights");
    B16Plan pl;
    std::string err;
    if (!make_b16_plan(*d, &pl, &err)) return api_fail(UNETMI_E_BAD_DESC, err);
    const B16Args& a = pl.args;
    std::memset(packed, 0, pl.packed_bytes);
    const int C = d->C0 + d->C1, K = d->K, KH = d->kh, KW = d->kw, G = a.G0 + a.G1, MT = pl.MT;
    const bool f16 = (d->reserved[2] & UNETMI_B16_IN_F16) != 0;
    for (int kc = 0; kc < pl.kchunks; ++kc)            // one image per 128-channel output chunk
    for (int q = 0; q < a.nphase; ++q) {
        const B16Phase& p = a.ph[q];
        unsigned short* img = reinterpret_cast<unsigned short*>(static_cast<char*>(packed) + (size_t)kc * a.w_chunk_bytes + p.w_off);
        const int ncp = a.pair ? 2 : 1;
        const int ntws = a.pair ? p.ntw[0] + p.ntw[1] : p.ntw[0];
        for (int g = 0; g < G; ++g)
            for (int th = 0; th < p.nth; ++th)
                for (int cp = 0; cp < ncp; ++cp)
                    for (int tw = 0; tw < p.ntw[cp]; ++tw) {
                        const int ky = pl.kh0[q] + th * pl.khs, kx = pl.kw0[q][cp] + tw * pl.kws;
                        for (int h = 0; h < a.hl; ++h)
                            for (int mt = 0; mt < MT; ++mt)
                                for (int hf = 0; hf < 2; ++hf)
                                    for (int row = 0; row < 32; ++row)
                                        for (int k = 0; k < 8; ++k) {
                                            const size_t tap = ((size_t)(g * a.hl + h) * p.nth + th) * ntws + (cp ? p.ntw[0] : 0) + tw;
                                            const size_t dst = ((((tap * MT + mt) * 2 + hf) * 32 + row) * 8) + k;
                                            const int cl = (g < a.G0 ? g : g - a.G0) * 16 + hf * 8 + k;
                                            const int ci = g < a.G0 ? (cl < d->C0 ? cl : -1) : (cl < d->C1 ? d->C0 + cl : -1);
                                            const int co = kc * 128 + mt * 32 + row;
                                            if (ci < 0 || co >= K || mt * 32 + row >= a.K) continue;
                                            const size_t src = d->transposed ? (((size_t)ci * K + co) * KH + ky) * KW + kx
                                                                             : (((size_t)co * C + ci) * KH + ky) * KW + kx;
                                            if (f16) {
                                                const unsigned short hi = host_f16(w[src]);
                                                img[dst] = h == 0 ? hi : host_f16(w[src] - host_f16_to_f32(hi));
                                            } else {
                                                const unsigned short hi = host_bf16(w[src]);
                                                img[dst] = h == 0 ? hi : host_bf16(w[src] - host_bf16_to_f32(hi));
                                            }
                                        }
                    }
    }
    return UNETMI_OK;
}

static int conv2d_b16_launch(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                             const float* bias, const float* scale, const float* shift, const float* act_param,
                             const float* gate, const float* head_w, const float* head_b, int head_P, int head_act,
                             int head_il, void* out, void* stream, const float* residual = nullptr) {
    if (!d) return api_fail(UNETMI_E_BAD_ARG, "null descriptor");
    B16Plan pl;
    std::string err;
    if (!make_b16_plan(*d, &pl, &err, head_P)) return api_fail(UNETMI_E_BAD_DESC, err);
    if (d->N == 0) return UNETMI_OK;
    if (!in0 || !wpacked || !bias || !scale || !shift || !out) return api_fail(UNETMI_E_BAD_ARG, "null tensor pointer");
    if (d->in_mode == UNETMI_IN_CONCAT && !in1) return api_fail(UNETMI_E_BAD_ARG, "concat needs in1");
    if (d->act == UNETMI_ACT_LEAKY && !act_param) return api_fail(UNETMI_E_BAD_ARG, "leaky/prelu needs act_param");
    if ((d->has_gate != 0) != (gate != nullptr)) return api_fail(UNETMI_E_BAD_ARG, "gate pointer vs has_gate");
    B16Args& a = pl.args;
    a.in0 = static_cast<const char*>(in0); a.in1 = static_cast<const char*>(in1); a.w = static_cast<const char*>(wpacked);
    a.bias = bias; a.scale = scale; a.shift = shift; a.act_param = act_param; a.gate = gate; a.out = out;
    a.head_w = head_w; a.head_b = head_b; a.head_P = head_P; a.head_act = head_act; a.head_il = head_il;
    a.residual = residual;
    if (residual && !a.out_nchw_f32) return api_fail(UNETMI_E_BAD_DESC, "conv2d_b16: a residual goes with the fp32 NCHW output only");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipErrorInvalidConfiguration;
    const bool gen = d->act != UNETMI_ACT_NONE && d->act != UNETMI_ACT_LEAKY && d->act != UNETMI_ACT_RELU;
    if (gen && (pl.MT > 1 || pl.args.pair))
        return api_fail(UNETMI_E_UNSUPPORTED, "conv2d_b16: SiLU / Softplus / Sigmoid / Tanh / GELU epilogues are built for K <= 32 "
                                              "and not for time-stride-2 transposed layers");
    const bool f16 = (d->reserved[2] & UNETMI_B16_IN_F16) != 0;
#define UNETMI_B16_RT2(MT_, F_)                                                                             \
    if (pl.RT == 2 && !pl.args.pair && pl.MT == MT_ && f16 == F_) {                                         \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel_rt2<MT_, F_>, &raised, pl, st); }       \
    if (pl.RT == 2 && pl.args.pair && pl.MT == MT_ && f16 == F_) {                                          \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel_pair_rt2<MT_, F_>, &raised, pl, st); }
#define UNETMI_B16(MT_, NT_, GEN_, F_)                                                                      \
    if (pl.RT == 1 && !pl.args.pair && pl.MT == MT_ && pl.NT == NT_ && gen == GEN_ && !pl.occ2 && f16 == F_) { \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel<MT_, NT_, GEN_, F_>, &raised, pl, st); }
#define UNETMI_B16_OCC2(MT_, NT_, F_)                                                                       \
    if (pl.RT == 1 && !pl.args.pair && pl.MT == MT_ && pl.NT == NT_ && !gen && pl.occ2 && f16 == F_) {      \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel_occ2<MT_, NT_, F_>, &raised, pl, st); }
#define UNETMI_B16_PAIR(MT_, F_)                                                                            \
    if (pl.RT == 1 && pl.args.pair && pl.MT == MT_ && !pl.occ2 && f16 == F_) {                              \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel_pair<MT_, F_>, &raised, pl, st); }      \
    if (pl.RT == 1 && pl.args.pair && pl.MT == MT_ && pl.occ2 && f16 == F_) {                               \
        static PerDeviceLatch raised; e = launch_b16(&conv_b16_kernel_pair_occ2<MT_, F_>, &raised, pl, st); }
#define UNETMI_B16_ALL(F_)                                                                                  \
    UNETMI_B16(1, 1, false, F_) UNETMI_B16(1, 2, false, F_) UNETMI_B16(2, 1, false, F_) UNETMI_B16(2, 2, false, F_) \
    UNETMI_B16(3, 1, false, F_) UNETMI_B16(3, 2, false, F_) UNETMI_B16(4, 1, false, F_)                     \
    UNETMI_B16(1, 1, true, F_) UNETMI_B16(1, 2, true, F_)                                                   \
    UNETMI_B16_OCC2(1, 1, F_) UNETMI_B16_OCC2(1, 2, F_) UNETMI_B16_OCC2(2, 1, F_) UNETMI_B16_OCC2(2, 2, F_) \
    UNETMI_B16_OCC2(3, 1, F_) UNETMI_B16_OCC2(3, 2, F_) UNETMI_B16_OCC2(4, 1, F_)                           \
    UNETMI_B16_PAIR(1, F_) UNETMI_B16_PAIR(2, F_) UNETMI_B16_PAIR(3, F_)                                    \
    UNETMI_B16_RT2(1, F_) UNETMI_B16_RT2(2, F_) UNETMI_B16_RT2(3, F_)
    UNETMI_B16_ALL(false)
    UNETMI_B16_ALL(true)
#undef UNETMI_B16_ALL
#undef UNETMI_B16_RT2
#undef UNETMI_B16_OCC2
#undef UNETMI_B16_PAIR
#undef UNETMI_B16
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("conv2d_b16 launch: ") + hipGetErrorString(e));
    return UNETMI_OK;
}

int unetmi_conv2d_b16_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                          const float* bias, const float* scale, const float* shift, const float* act_param,
                          const float* gate, void* out, void* stream) {
    return conv2d_b16_launch(d, in0, in1, wpacked, bias, scale, shift, act_param, gate, nullptr, nullptr, 0,
                             UNETMI_ACT_NONE, 0, out, stream);
}

int unetmi_conv2d_b16_nchw_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                               const float* bias, const float* scale, const float* shift, const float* act_param,
                               const float* gate, const float* residual, float* out, void* stream) {
    if (!d) return api_fail(UNETMI_E_BAD_ARG, "null descriptor");
    UnetmiConvDesc dd = *d;
    dd.reserved[2] = (dd.reserved[2] & UNETMI_B16_IN_F16) | UNETMI_B16_OUT_NCHW_F32;
    return conv2d_b16_launch(&dd, in0, in1, wpacked, bias, scale, shift, act_param, gate, nullptr, nullptr, 0,
                             UNETMI_ACT_NONE, 0, out, stream, residual);
}

int unetmi_conv2d_b16_head_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                               const float* bias, const float* scale, const float* shift, const float* act_param,
                               const float* gate, const float* head_w, const float* head_b, int head_channels,
                               int head_act, int interleaved, float* out, void* stream) {
    if (head_channels < 1 || head_channels > kB16MaxHead)
        return api_fail(UNETMI_E_UNSUPPORTED, "conv2d_b16_head: 1 to 4 head channels");
    if (!head_w) return api_fail(UNETMI_E_BAD_ARG, "conv2d_b16_head: null head weights");
    if (d && d->act != UNETMI_ACT_NONE && d->act != UNETMI_ACT_LEAKY && d->act != UNETMI_ACT_RELU)
        return api_fail(UNETMI_E_UNSUPPORTED, "conv2d_b16_head: the layer's own activation must be none, LeakyReLU / PReLU or ReLU");
    if (head_act < UNETMI_ACT_NONE || head_act >= UNETMI_ACT_GELU || head_act == UNETMI_ACT_LEAKY)
        return api_fail(UNETMI_E_BAD_DESC, "conv2d_b16_head: head activation must be none, ReLU, SiLU, Softplus, Sigmoid or Tanh");
    return conv2d_b16_launch(d, in0, in1, wpacked, bias, scale, shift, act_param, gate, head_w, head_b, head_channels,
                             head_act, interleaved ? 1 : 0, out, stream);
}

int unetmi_blocked_from_nchw(int N, int C, int HW, const void* src, int src_dtype, void* dst, void* stream) {
    return unetmi_blocked_from_nchw_sum(N, C, HW, src, nullptr, src_dtype, dst, UNETMI_DTYPE_BF16, stream);
}

int unetmi_blocked_from_nchw_sum(int N, int C, int HW, const void* src, const void* src2, int src_dtype, void* dst,
                                 int dst_dtype, void* stream) {
    if (dst_dtype != UNETMI_DTYPE_BF16 && dst_dtype != UNETMI_DTYPE_F16)
        return api_fail(UNETMI_E_BAD_DESC, "blocked_from_nchw: the blocked tensor is bfloat16 or fp16");
    const int dst_f16 = dst_dtype == UNETMI_DTYPE_F16;
    if (N < 0 || C < 1 || HW < 1) return api_fail(UNETMI_E_BAD_DESC, "blocked_from_nchw: bad sizes");
    if (N == 0) return UNETMI_OK;
    if (!src || !dst) return api_fail(UNETMI_E_BAD_ARG, "blocked_from_nchw: null pointer");
    const int CB = 2 * ((C + 15) / 16);
    if (N > 65535 || CB > 65535) return api_fail(UNETMI_E_UNSUPPORTED, "blocked_from_nchw: N or C too large");
    const dim3 grid((HW + 1023) / 1024, CB, N);              // four pixels per thread
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (src_dtype == UNETMI_DTYPE_BF16) hipLaunchKernelGGL((to_blocked_kernel<true>), grid, dim3(256), 0, st, src, src2, static_cast<char*>(dst), C, CB, HW, dst_f16);
    else if (src_dtype == UNETMI_DTYPE_F32) hipLaunchKernelGGL((to_blocked_kernel<false>), grid, dim3(256), 0, st, src, src2, static_cast<char*>(dst), C, CB, HW, dst_f16);
    else return api_fail(UNETMI_E_BAD_DESC, "blocked_from_nchw: unknown dtype");
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("blocked_from_nchw launch: ") + hipGetErrorString(e));
}

int unetmi_nchw_from_blocked(int N, int C, int HW, const void* src, int src_dtype, void* dst, int dst_dtype, void* stream) {
    if (src_dtype != UNETMI_DTYPE_BF16 && src_dtype != UNETMI_DTYPE_F16)
        return api_fail(UNETMI_E_BAD_DESC, "nchw_from_blocked: the blocked tensor is bfloat16 or fp16");
    const int src_f16 = src_dtype == UNETMI_DTYPE_F16;
    if (N < 0 || C < 1 || HW < 1) return api_fail(UNETMI_E_BAD_DESC, "nchw_from_blocked: bad sizes");
    if (N == 0) return UNETMI_OK;
    if (!src || !dst) return api_fail(UNETMI_E_BAD_ARG, "nchw_from_blocked: null pointer");
    const int CB = 2 * ((C + 15) / 16);
    if (N > 65535 || CB > 65535) return api_fail(UNETMI_E_UNSUPPORTED, "nchw_from_blocked: N or C too large");
    const dim3 grid((HW + 255) / 256, CB, N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dst_dtype == UNETMI_DTYPE_BF16) hipLaunchKernelGGL((from_blocked_kernel<true>), grid, dim3(256), 0, st, static_cast<const char*>(src), dst, C, CB, HW, src_f16);
    else if (dst_dtype == UNETMI_DTYPE_F32) hipLaunchKernelGGL((from_blocked_kernel<false>), grid, dim3(256), 0, st, static_cast<const char*>(src), dst, C, CB, HW, src_f16);
    else return api_fail(UNETMI_E_BAD_DESC, "nchw_from_blocked: unknown dtype");
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("nchw_from_blocked launch: ") + hipGetErrorString(e));
}

}  // extern "C"
