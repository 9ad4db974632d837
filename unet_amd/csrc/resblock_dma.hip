// One layer of the residual block (UNet_m.py:545-561) for a single input, as a software
// pipeline in which every wave carries its share of ALL stages inside its own matrix-instruction stream:
//
//     out = [x +] BN(PReLU(pointwise_1x1(depthwise_3x3_dilated(x))))          64 channels
//
// The layer moves one read and one write of the activation.  What bounds it on this chip is the number of
// read requests a compute unit's vector cache keeps outstanding (measured: ~60 requests at ~800 cycles each),
// so the kernel reads every input byte ONCE, with 16 bytes per lane, and never asks for the same data twice:
//   * the input patch travels by LDS-DMA (buffer_load_dwordx4 ... lds) through a RING of patch buffers that
//     runs several steps ahead of its use and is waited for with counted vmcnt, never a drain;
//   * the running-add operand (the centre of the patch) is picked out of that same LDS image into registers
//     as the patch passes by, instead of being read from memory again for the epilogue.
//
// A workgroup (4 waves) walks tiles of 4 rows x 64 time steps; a tile is 16 steps of 4 input channels, wave w
// owning channel w of a step's four (it fetches that channel's patch, 6 rows x (64 + 2d) columns, and computes
// its depthwise 3x3) and image row w of the tile (its 64 pixels x 64 output channels of the pointwise product).
// In step s a wave
//   * issues the DMA of its channel's patch of step s+RING-1 into the ring buffer step s-1 was read from,
//   * multiplies the depthwise result of step s with the 64x64 pointwise weights (resident in LDS in operand
//     order) on v_mfma_f32_32x32x2_f32 (8 per step), and between those MFMAs computes the depthwise of step s+1
//     on the vector ALUs (lane = time step, four output rows from six patch rows) into the other B-operand
//     buffer, and copies the running-add values of step s out of its patch,
// then waits until its piece of the patch of step s+2 has landed (younger DMA groups stay in flight) and meets
// the other waves at ONE barrier.  Steps run on across tiles; the epilogue (PReLU, BatchNorm affine, running
// add, store: one instruction = 64 consecutive pixels of one channel) follows the barrier of a tile's last step.
//
// Image borders: rows outside the image and 4-column groups that lie wholly outside it are range-check misses
// of the DMA (zeros).  The left halo is padded to whole groups, so only the right border can cut a group: it then
// brings in the first values of the next row (or, at the very end of the sample, per-dword range-check zeros),
// which the wave that owns the channel overwrites with zeros in LDS before its depthwise reads them (border
// tiles only).
//
// vmcnt bookkeeping: every wave issues the SAME vector-memory instructions in the same order (a step past the
// last tile and a row outside the image issue theirs against an empty descriptor or an out-of-range offset),
// so "all but the N youngest" names the same instructions in every wave.
//
// Workgroup -> tile order: when the batch is a multiple of 8, the workgroups of one XCD (blockIdx.x mod 8)
// work through whole samples together, so neighbouring tiles' shared halo meets in that XCD's L2.
//
// Storage: fp32, or bfloat16 (a 16-byte DMA group then holds 8 activations; they are widened when the depthwise
// and the running-add copy read them out of LDS, and the result is rounded once on store); arithmetic is fp32
// either way.  The register-staged kernel in resblock.hip serves what this one does not: the input addend of a
// decoder block's first layer.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "io_types.hpp"
#include "resblock_common.hpp"

// Timing-only builds (tools/rb_bench.hip) switch stages off with -DRB_ABLATE=<bits> to price them; the library is
// built without it.
#ifndef RB_ABLATE
#define RB_ABLATE 0
#endif
#define RB_OFF(bit) ((RB_ABLATE & (bit)) != 0)

namespace unetmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_void_ptr_rb;

template <int D, bool BF16>
struct RbGeom {
    static constexpr int EB = BF16 ? 2 : 4;                 // bytes of a stored activation
    static constexpr int GE = 16 / EB;                      // activations of one 16-byte DMA group
    static constexpr int TC = 64, TR = kRbTR, CK = 4, NCH = kRbC / CK;
    static constexpr int PROWS = TR + 2;
    static constexpr int DL = (D + GE - 1) / GE * GE;       // left halo rounded up to whole groups, so that no group
                                                            // straddles the left border (tiles start at multiples of 64)
    static constexpr int TCP = (DL + TC + D + GE - 1) / GE * GE;   // patch columns; column c = image column col0 - DL + c
    static constexpr int QPR = TCP / GE;                    // 16-byte groups per patch row
    static constexpr int NQ = PROWS * QPR;                  // groups (= DMA lanes) of one channel's patch
    static constexpr int NDI = (NQ + 63) / 64;              // DMA instructions per wave and step
    static constexpr int CHB = PROWS * TCP * EB;            // bytes of one channel's patch
    static constexpr int IN_BYTES = CK * CHB;
#ifdef RB_RING
    static constexpr int RING = RB_RING;
#else
    // patch buffers; the DMA runs RING-1 steps ahead of the depthwise.  Measured at 3 .. 6: no gain beyond 4 -- the
    // vector cache's outstanding-request limit, not the prefetch distance, sets the rate.
    static constexpr int RING = 4;
#endif
    static constexpr int YOUNG = (RING - 3) * NDI;          // DMA instructions younger than the patch a step's wait is for
    static constexpr int BP = TR * TC + 16;                 // B-operand pitch of a channel (lane halves in different banks)
    static constexpr int B_FLOATS = CK * BP;
    static constexpr int DWP = 12;                          // depthwise taps of a channel: 9 + bias, padded to 16 bytes
    static constexpr int W_FLOATS = kRbC * kRbC, PAR_FLOATS = 3 * kRbC;
    static constexpr size_t LDS =
        (size_t)(W_FLOATS + 2 * B_FLOATS + PAR_FLOATS + kRbC * DWP) * sizeof(float) + (size_t)RING * IN_BYTES;
    static constexpr int kStores = 64;                      // store instructions per wave and tile
};

// activation as stored (fp32, or bfloat16 bits widened exactly) -> float
template <bool BF16> struct RbAct;
template <> struct RbAct<false> {
    typedef float T;
    static __device__ __forceinline__ float widen(float v) { return v; }
};
template <> struct RbAct<true> {
    typedef unsigned short T;
    static __device__ __forceinline__ float widen(unsigned short v) { return __builtin_bit_cast(float, (unsigned)v << 16); }
};

constexpr int kRbSentinel = 0x60000000;      // byte offset beyond any sample this kernel accepts (< 0x20000000 bytes)

// f(integral_constant<int, I>) for I = FROM .. TO-1, unrolled with I a compile-time value
template <int FROM, int TO, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (FROM < TO) {
        f(std::integral_constant<int, FROM>{});
        static_for<FROM + 1, TO>(f);
    }
}

// 16 bytes per lane from a buffer straight into LDS (lane l lands at lds + 16*l); out-of-range dwords arrive as 0.
// A plain device function: the 16-byte form only exists for this target, which a kernel template's host-side
// instantiation cannot see.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, void* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr_rb)lds, 16, voff, soff, 0, 0);
}

// all but the N youngest vector-memory instructions of this wave, and all of its LDS instructions, have completed
template <int N>
__device__ __forceinline__ void wait_vm_and_lds() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N < 63 ? N : 63) : "memory");
}

template <int D, bool BF16>
__global__ __launch_bounds__(256, 2) void resblock_dma_kernel(const ResblockArgs a) {
    using G = RbGeom<D, BF16>;
    using Act = RbAct<BF16>;
    typedef typename Act::T act_t;
    constexpr int TCP = G::TCP, BP = G::BP, RING = G::RING, NCH = G::NCH, EB = G::EB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_w = smem;                                      // [32 k-steps][2 tiles][64 lanes] weight operands
    float* s_b = s_w + G::W_FLOATS;                         // [2][4][BP] depthwise results = pixel operands
    float* s_par = s_b + 2 * G::B_FLOATS;                   // [3][64] pointwise bias, BatchNorm scale, shift
    float* s_dw = s_par + G::PAR_FLOATS;                    // [64][12] depthwise taps + bias
    char* s_in = reinterpret_cast<char*>(s_dw + kRbC * G::DWP);   // [RING][4 channels][6][TCP] input patches, as stored

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int HW = a.H * a.W;

    for (int i = tid; i < kRbC * kRbC; i += 256) {
        const int l = i & 63, t = (i >> 6) & 1, ks = i >> 7;
        s_w[i] = a.pw_w[(t * 32 + (l & 31)) * kRbC + 2 * ks + (l >> 5)];
    }
    if (tid < kRbC) {
        s_par[tid] = a.pw_b ? a.pw_b[tid] : 0.f;
        s_par[kRbC + tid] = a.scale[tid];
        s_par[2 * kRbC + tid] = a.shift[tid];
    }
    for (int i = tid; i < kRbC * G::DWP; i += 256) {
        const int ch = i / G::DWP, k = i - ch * G::DWP;
        s_dw[i] = k < 9 ? a.dw_w[ch * 9 + k] : (k == 9 && a.dw_b ? a.dw_b[ch] : 0.f);
    }
    const float alpha_m1 = a.alpha[0] - 1.f;

    // ---- which tiles are this workgroup's ----
    const int tps = a.tiles_h * a.tiles_w;                  // tiles per sample
    const bool by_xcd = !RB_OFF(64) && (a.N & 7) == 0 && (gridDim.x & 7) == 0;
    const int t_first = by_xcd ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int t_count = by_xcd ? (a.N >> 3) * tps : a.ntiles;
    auto tile_coords = [&](int u, int& n, int& row0, int& col0) {
        n = u / tps;
        const int rem = u - n * tps;
        if (by_xcd) n = (n << 3) + (int)(blockIdx.x & 7);
        row0 = (rem / a.tiles_w) * G::TR;
        col0 = (rem % a.tiles_w) * G::TC;
    };
    int tile = t_first;
    if (tile >= t_count) return;

    // ---- DMA geometry of this lane: group q = i*64 + lane of its wave's channel patch -> (patch row, first column).
    // toff[i]: byte offset of the group inside a channel plane, or an offset beyond the buffer (the DMA writes 0)
    // for a row outside the image and for a group wholly outside it. ----
    int toff[G::NDI];
    auto aim = [&](int row0, int col0) {
#pragma unroll
        for (int i = 0; i < G::NDI; ++i) {
            const int q = i * 64 + lane;
            const int r = q / G::QPR, qc = q - r * G::QPR;
            const int grow = row0 - 1 + r, gc = col0 - G::DL + G::GE * qc;
            bool ok = (unsigned)grow < (unsigned)a.H && gc + G::GE - 1 >= 0 && gc < a.W;
            if (RB_OFF(128) && (r == 0 || r == G::PROWS - 1)) ok = false;                    // timing: no halo rows
            if (RB_OFF(256) && (gc < col0 || gc + G::GE > col0 + G::TC)) ok = false;         // timing: no halo columns
            toff[i] = ok ? (grow * a.W + gc) * EB : kRbSentinel;
        }
    };
    // this wave's channel of (sample n, chunk) -> ring buffer buf; live = false: no such step (empty descriptor)
    auto dma = [&](int n, int chunk, int buf, bool live) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(a.x) + (size_t)n * kRbC * HW * EB), 0, live ? kRbC * HW * EB : 0,
            0x00020000);
        char* dst = s_in + buf * G::IN_BYTES + wave * G::CHB;
        const int soff = (chunk * G::CK + wave) * HW * EB;
#pragma unroll
        for (int i = 0; i < G::NDI; ++i) {
            if ((i + 1) * 64 <= G::NQ || i * 64 + lane < G::NQ)      // the last instruction may be partly masked
                dma16(rs, dst + i * 1024, toff[i], soff);
        }
    };
    // zeros over the columns of this wave's channel patch that lie left or right of the image (border tiles)
    auto fix_borders = [&](act_t* pch, int col0) {
        const int nleft = G::DL - col0;                      // columns [0, nleft) are left of the image
        if (nleft > 0) {
            const int c = lane & 15;
#pragma unroll
            for (int r0 = 0; r0 < G::PROWS; r0 += 4) {
                const int r = r0 + (lane >> 4);
                if (r < G::PROWS && c < nleft) pch[r * TCP + c] = 0;
            }
        }
        const int cr0 = a.W - col0 + G::DL;                 // columns [cr0, TCP) are right of the image
        if (cr0 < TCP) {
            const int c = cr0 + (lane & 31);
#pragma unroll
            for (int r0 = 0; r0 < G::PROWS; r0 += 2) {
                const int r = r0 + (lane >> 5);
                if (c < TCP) pch[r * TCP + c] = 0;
            }
        }
    };

    // depthwise 3x3, dilation (1, D), of this wave's channel of a chunk (global channel chg): ring buffer -> B buffer
    struct DwIn { float k[9], bias, v[G::PROWS][3]; };
    auto dw_load = [&](const act_t* pch, int chg, DwIn& d) {
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(s_dw + chg * G::DWP);
        const f32x4 k1 = *reinterpret_cast<const f32x4*>(s_dw + chg * G::DWP + 4);
        const f32x2 k2 = *reinterpret_cast<const f32x2*>(s_dw + chg * G::DWP + 8);
        d.k[0] = k0[0]; d.k[1] = k0[1]; d.k[2] = k0[2]; d.k[3] = k0[3];
        d.k[4] = k1[0]; d.k[5] = k1[1]; d.k[6] = k1[2]; d.k[7] = k1[3];
        d.k[8] = k2[0]; d.bias = k2[1];
#pragma unroll
        for (int r = 0; r < G::PROWS; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) d.v[r][q] = Act::widen(pch[r * TCP + lane + (G::DL - D) + q * D]);
    };
    auto dw_row = [&](const DwIn& d, float* pb, int r) {
        float o = d.bias;
#pragma unroll
        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
            for (int q = 0; q < 3; ++q) o = fmaf(d.v[r + kr][q], d.k[kr * 3 + q], o);
        pb[wave * BP + r * G::TC + lane] = o;
    };

    // accumulators [output-channel tile t][32-pixel block h]: lane l holds pixel h*32 + (l&31) of image row
    // row0 + wave, register r channel t*32 + (r&3) + 8*(r>>2) + 4*(l>>5); res: the running-add operand, same layout
    f32x16 acc[2][2];
    float res[2][2][16];
    auto reset_acc = [&]() {               // accumulators start from the pointwise bias of their output channel
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pb = s_par[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
                acc[t][0][r] = pb;
                acc[t][1][r] = pb;
            }
    };

    int n, row0, col0;                         // tile the MFMAs are on
    int dn, drow0, dcol0;                      // tile the DMA is on
    tile_coords(tile, n, row0, col0);
    dn = n; drow0 = row0; dcol0 = col0;
    aim(drow0, dcol0);
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) dma(dn, s, s, true);          // patches of steps 0 .. RING-2
    wait_vm_and_lds<0>();
    __syncthreads();                           // staged weights, parameters and taps; the first patches landed
    reset_acc();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 16; ++r) res[t][h][r] = 0.f;
    {
        DwIn d;
        act_t* pch = reinterpret_cast<act_t*>(s_in + wave * G::CHB);
        fix_borders(pch, col0);
        dw_load(pch, wave, d);
#pragma unroll
        for (int r = 0; r < G::TR; ++r) dw_row(d, s_b, r);
    }
    __syncthreads();

    // ring positions: buffer of the patch of this step (cur_rb), the DMA of this step writes the one before it
    int cur_rb = 0;
    bool dlive = true;                         // the tile the DMA is on exists
    while (true) {
        const int next_tile = tile + t_step;
        const bool has_next = next_tile < t_count;
        static_for<0, NCH>([&](auto p_tag) {
            constexpr int p = decltype(p_tag)::value;
            constexpr int cur = p & 1;
            // ---- DMA of step + RING-1, into the ring buffer of step - 1 ----
            if (p == NCH - (RING - 1)) {       // from here on the DMA works on the next tile
                dlive = has_next;
                if (has_next) {
                    tile_coords(next_tile, dn, drow0, dcol0);
                    aim(drow0, dcol0);
                }
            }
            int prev_rb = cur_rb - 1;
            if (prev_rb < 0) prev_rb = RING - 1;
            if (!RB_OFF(2)) dma(dn, (p + RING - 1) & (NCH - 1), prev_rb, dlive);
            int next_rb = cur_rb + 1;
            if (next_rb == RING) next_rb = 0;

            // ---- operands of this step's MFMAs, inputs of the next step's depthwise ----
            const float* sb = s_b + cur * G::B_FLOATS + wave * G::TC + j;
            const float* sw = s_w + p * (G::CK / 2) * 128 + lane;
            float pv[2][2], wv[2][2];          // [k-step][pixel block] / [k-step][channel tile]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pv[ks][h] = sb[(2 * ks + half) * BP + h * 32];
                    wv[ks][h] = sw[(ks * 2 + h) * 64];
                }
            DwIn d;
            act_t* pch = reinterpret_cast<act_t*>(s_in + next_rb * G::IN_BYTES + wave * G::CHB);
            if (!RB_OFF(1)) {
                // On the very last step of the workgroup there is no next step: the depthwise then runs on a
                // stale patch into a B buffer nobody reads, which costs less than a second code path.
                const int ncol0 = p == NCH - 1 ? dcol0 : col0;
                if (ncol0 < G::DL || ncol0 + G::TC + D > a.W) fix_borders(pch, ncol0);
                dw_load(pch, ((p + 1) & (NCH - 1)) * G::CK + wave, d);
            }
            // ---- the running-add operand of this step's four channels: centre of its patch.  They belong to the
            // lanes of one half (channel = 4*(l>>5) + 8*(r>>2) + (r&3) within a tile of 32). ----
            if (a.residual && half == (p & 1)) {
                const act_t* pc = reinterpret_cast<const act_t*>(s_in + cur_rb * G::IN_BYTES) + (wave + 1) * TCP + G::DL + j;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        res[p >> 3][h][4 * ((p & 7) >> 1) + i] = Act::widen(pc[i * (G::CHB / EB) + h * 32]);
            }
            float* pbn = s_b + (cur ^ 1) * G::B_FLOATS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (!RB_OFF(16))
                            acc[t][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ks][t], pv[ks][h], acc[t][h], 0, 0, 0);
                    if (!RB_OFF(1)) dw_row(d, pbn, ks * 2 + t);
                }
            // ---- this wave's piece of the patch of step + 2 has landed; younger DMA groups (and, at the start
            // of a tile, the previous tile's stores) stay in flight ----
            constexpr bool after_stores = p < RING - 3;
            if (RB_OFF(32)) wait_vm_and_lds<0>();
            else wait_vm_and_lds<G::YOUNG + (after_stores ? G::kStores : 0)>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur_rb = next_rb;
        });

        // ---- epilogue: PReLU (v + (alpha - 1) * min(v, 0)), BatchNorm affine, running add, store ----
        {
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(a.out) + (size_t)n * kRbC * HW * EB, 0,
                                                                  RB_OFF(8) ? 0 : kRbC * HW * EB, 0x00020000);
            const int orow = row0 + wave;
            // one store instruction = 64 consecutive pixels of ONE channel (256 contiguous bytes): the lower lanes'
            // pixels 32..63 and the upper lanes' pixels 0..31 change places first
            const int ocol64 = col0 + lane;
            const int obase64 = (orow < a.H && ocol64 < a.W) ? (orow * a.W + ocol64) * EB : kRbSentinel;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cou = t * 32 + (r & 3) + 8 * (r >> 2);
                    const float sc = s_par[kRbC + cou + 4 * half], sh = s_par[2 * kRbC + cou + 4 * half];
                    float v[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float x = acc[t][h][r];
                        x = fmaf(fminf(x, 0.f), alpha_m1, x);
                        v[h] = fmaf(x, sc, sh) + res[t][h][r];
                    }
                    unsigned s0 = __builtin_bit_cast(unsigned, v[0]), s1 = __builtin_bit_cast(unsigned, v[1]);
                    swap_lane_halves(s0, s1);          // s0: channel cou, s1: channel cou + 4, pixel = lane in both
                    buf_st_act<BF16>(__builtin_bit_cast(float, s0), rs_out, obase64, cou * HW * EB);
                    buf_st_act<BF16>(__builtin_bit_cast(float, s1), rs_out, obase64, (cou + 4) * HW * EB);
                }
        }
        if (!has_next) break;
        tile = next_tile;
        n = dn; row0 = drow0; col0 = dcol0;
        reset_acc();
    }
}

template <int D, bool BF16>
static hipError_t launch_dma_t(hipStream_t s, const ResblockArgs& a, int grid) {
    static PerDeviceLatch raised;                   // dynamic LDS beyond 64 KB: once per kernel and device
    const hipError_t e = raised.once([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_dma_kernel<D, BF16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)RbGeom<D, BF16>::LDS);
    });
    if (e != hipSuccess) return e;
    constexpr size_t lds = RbGeom<D, BF16>::LDS;
    hipLaunchKernelGGL((resblock_dma_kernel<D, BF16>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_resblock_dma(int dilation_w, hipStream_t s, ResblockArgs a, bool bf16) {
    a.tiles_h = (a.H + kRbTR - 1) / kRbTR;
    a.tiles_w = (a.W + 63) / 64;
    const long long nt = (long long)a.N * a.tiles_h * a.tiles_w;
    if (nt < 1 || nt > 0x7fffffffll) return hipErrorInvalidValue;
    a.ntiles = (int)nt;
    // persistent: two workgroups per compute unit, each walks its tiles with the pipeline running across them
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = (int)(nt < 2ll * cus ? nt : 2ll * cus);
    switch (dilation_w) {
        case 1: return bf16 ? launch_dma_t<1, true>(s, a, grid) : launch_dma_t<1, false>(s, a, grid);
        case 2: return bf16 ? launch_dma_t<2, true>(s, a, grid) : launch_dma_t<2, false>(s, a, grid);
        case 4: return bf16 ? launch_dma_t<4, true>(s, a, grid) : launch_dma_t<4, false>(s, a, grid);
        case 8: return bf16 ? launch_dma_t<8, true>(s, a, grid) : launch_dma_t<8, false>(s, a, grid);
        case 16: return bf16 ? launch_dma_t<16, true>(s, a, grid) : launch_dma_t<16, false>(s, a, grid);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace unetmi
