// The main convolution kernel: same tiling, MFMA loop and epilogue as
// conv_mfma_kernel (conv_kernel.hpp), but the input patch and the weights reach LDS
// by LDS-DMA (buffer_load ... lds: no data registers, no ds_write pass) into a
// DOUBLE-buffered image, and the DMA instructions for chunk c+1 are issued one per
// tap from inside the MFMA loop of chunk c.
//
// Why: on gfx950 a wave that is issuing back-to-back MFMAs starves the other wave on
// its SIMD (a burst of 32 loads beside it took 67,000 cycles instead of 800), so
// staging work placed in a phase of its own does not overlap with the partner's
// matrix work the way wave-level multithreading suggests.  Work that a wave
// interleaves with its OWN MFMAs, on the other hand, issues in their shadow.  So
// each wave carries its share of the staging inside its MFMA stream and the only
// non-matrix phases left are the prologue, one barrier per chunk and the epilogue.
//
// Out-of-image lanes, channels past the end and unused slots are buffer-range
// misses: the DMA writes 0 for them.  A chunk never mixes the two concat sources
// (the launcher sends such layers, and the skip-add, to conv_mfma_kernel).
#pragma once
#include "conv_kernel.hpp"

namespace unetmi {

typedef __attribute__((address_space(3))) void* lds_void_ptr;

// Staging work a wave does between two taps of its MFMA loop: at most ONE LDS-DMA
// instruction (its 64-lane piece of input slot q, then of weight slot q - nin_slots).
// Everything except the lane's (c, r, cl) walk is wave-uniform and stays in SGPRs.
// NT = threads of the workgroup = elements one slot moves.
template <int NT>
struct DmaSide {
    int q, nin_slots, nslots;                  // next slot to issue / input slots / all slots
    const float* in_base; int in_bytes;        // the chunk's input channels (buffer range = zero fill)
    const float* w_base; int w_bytes;          // the chunk's packed weights
    float* dst_in; float* dst_w;               // LDS images being filled (+ this wave's piece)
    int row0, col0, H, W, HW, ncols, nrows, step_c, step_r, step_cl, tid16;
    int c, r, cl;                              // this lane's column / row / channel for input slot q

    __device__ __forceinline__ bool pending() const { return __builtin_amdgcn_readfirstlane(q) < nslots; }

    __device__ __forceinline__ void operator()() {
        const int qq = __builtin_amdgcn_readfirstlane(q);
        if (qq < nin_slots) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_base), 0, in_bytes, 0x00020000);
            const int grow = row0 + r, gcol = col0 + c;
            const bool ok = (unsigned)grow < (unsigned)H && (unsigned)gcol < (unsigned)W;
            const int voff = ok ? (cl * HW + grow * W + gcol) * 4 : (int)0x80000000;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(dst_in + qq * NT), 4, voff, 0, 0, 0);
            c += step_c; r += step_r; cl += step_cl;      // advance this lane by NT elements
            if (c >= ncols) { c -= ncols; ++r; }
            if (r >= nrows) { r -= nrows; ++cl; }
        } else if (qq < nslots) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w_base), 0, w_bytes, 0x00020000);
            const int ws = qq - nin_slots;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)(dst_w + ws * (4 * NT)), 16,
                                                     ws * (16 * NT) + tid16, 0, 0, 0);
        }
        q = qq + 1;
    }
};

template <int NG, int TPG, int WM, int WN, int NTW0, int NTW1>
__global__ __launch_bounds__(64 * WM * WN, (NG == 1 && TPG <= 4) ? 3 : 2) void conv_dma_kernel(const ConvArgs p) {
    constexpr int NT = 64 * WM * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int buf_floats = p.in_tile_floats + p.w_tile_floats;
    float* s_par = smem + 2 * buf_floats;          // [5][NB]: bias, scale, shift, slope, gate

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, j = lane & 31;

    // ---- which tile is this workgroup's (same numbering as conv_mfma_kernel) ----
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

    int ib[NG][TPG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int t = 0; t < TPG; ++t)
            ib[g][t] = half * p.plane + ((wm * TPG + t) * p.h_in_step) * p.ncols +
                       j * p.w_in_step + (p.w_base[g] - p.w_colmin);
    const int wlane = half * p.NB + wn * 32 + j;

    // ---- DMA bookkeeping -----------------------------------------------------------
    // Input slot s moves elements e = s*NT + tid of the flat [ci][row][col] image; a
    // wave's 64 lanes land at consecutive LDS dwords.  (c, r, cl) = this thread's
    // column / row / channel for the slot about to be issued, stepped by +NT elements.
    const int in_elems = p.ci_chunk * p.plane;
    const int nin_slots = (in_elems + NT - 1) / NT;
    const int nw_slots = ((chunk_w_floats >> 2) + NT - 1) / NT;
    int c0, r0, cl0;
    {
        const int rr = __umulhi((unsigned)tid, p.magic_ncols);
        c0 = tid - rr * p.ncols;
        cl0 = (p.nrows == 1) ? rr : (int)__umulhi((unsigned)rr, p.magic_nrows);
        r0 = rr - cl0 * p.nrows;
    }
    const int step_rr = NT / p.ncols, step_c = NT - step_rr * p.ncols;       // NT = step_rr*ncols + step_c
    const int step_cl = step_rr / p.nrows, step_r = step_rr - step_cl * p.nrows;
    const float* wsrc_base = p.wpacked + p.w_phase_off[a] + (size_t)cb * p.nchunk * chunk_w_floats;

    DmaSide<NT> side;
    side.nin_slots = nin_slots; side.nslots = nin_slots + nw_slots; side.q = side.nslots;
    side.row0 = row0; side.col0 = col0; side.H = p.H; side.W = p.W; side.HW = HW;
    side.ncols = p.ncols; side.nrows = p.nrows;
    side.step_c = step_c; side.step_r = step_r; side.step_cl = step_cl; side.tid16 = tid * 16;
    side.c = c0; side.r = r0; side.cl = cl0;

    auto open_chunk = [&](int chunk, int buf) {      // point the DMA state at (chunk -> LDS buffer buf)
        const bool live = chunk < p.nchunk && !(ABLATE(8) && chunk > 0);
        int ch;
        long long first;
        const float* base = p.in0;
        if (chunk >= p.nchunk0) {                    // second source of a concatenation
            const int ci_base = (chunk - p.nchunk0) * p.ci_chunk;
            base = p.in1; ch = min(p.ci_chunk, p.C1 - ci_base); first = (long long)n * p.C1 + ci_base;
        } else {
            const int ci_base = chunk * p.ci_chunk;
            ch = min(p.ci_chunk, p.C0 - ci_base); first = (long long)n * p.C0 + ci_base;
        }
        side.in_base = base + first * HW;
        side.in_bytes = live ? ch * HW * 4 : 0;
        side.w_base = wsrc_base + (size_t)(live ? chunk : 0) * chunk_w_floats;
        side.w_bytes = live ? chunk_w_floats * 4 : 0;
        side.dst_in = smem + buf * buf_floats + wave * 64;
        side.dst_w = smem + buf * buf_floats + p.in_tile_floats + wave * 256;
        side.c = c0; side.r = r0; side.cl = cl0;
        side.q = (live && !(ABLATE(16) && chunk > 0)) ? 0 : side.nslots;
    };

    // ---- prologue: epilogue parameters, first chunk ----------------------------------
    float par[5] = {0.f, 1.f, 0.f, 0.f, 1.f};
    if (tid < p.NB) {
        const int co = cb * p.NB + tid;
        if (co < p.K) {
            par[0] = p.bias[co]; par[1] = p.scale[co]; par[2] = p.shift[co];
            if (p.act_param) par[3] = p.act_param[p.act_pc ? co : 0];
            if (p.gate) par[4] = p.gate[(size_t)n * p.K + co];
        }
    }
    open_chunk(0, 0);
    for (int q = 0; q < nin_slots + nw_slots; ++q) side();
    if (tid < p.NB) {
#pragma unroll
        for (int q = 0; q < 5; ++q) s_par[q * p.NB + tid] = par[q];
    }
    __syncthreads();          // drains the DMA (vmcnt 0) and the parameter writes

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        const int cur = chunk & 1;
        const float* s_in = smem + cur * buf_floats;
        const float* s_w = s_in + p.in_tile_floats;
        open_chunk(chunk + 1, cur ^ 1);               // fetched from inside the MFMA loops below
        {
            const float* wp = s_w + wlane;
            if constexpr (NG == 2 && NTW0 > 0) {
                group_mfma2<TPG, NTW0, NTW1, 32 * WN>(acc[0], acc[1], s_in, wp, ib[0], p.w_base[1] - p.w_base[0],
                                                      cpc, nth, p, side);
            } else {
                group_mfma<TPG, NTW0, 32 * WN>(acc[0], s_in, wp, ib[0], cpc, nth, p.w_ntap[0], p, side);
                if (NG == 2) {
                    wp += cpc * nth * p.w_ntap[0] * 2 * p.NB;
                    group_mfma<TPG, NTW1, 32 * WN>(acc[NG - 1], s_in, wp, ib[NG - 1], cpc, nth, p.w_ntap[1], p,
                                                   side);
                }
            }
        }
        while (side.pending()) side();                // more slots than taps
        if (!ABLATE(32)) __syncthreads();      // next image landed (vmcnt 0) and everyone is done reading this one
    }

    if (p.stats)                   // workgroup-uniform
        conv_tile_stats<NG, TPG>(acc, p, s_par, smem, WM, cb, a, i0, j0, wm, wn, half, j, bid_all / p.nco_blk);
    conv_epilogue<NG, TPG>(acc, p, s_par, n, cb, a, i0, j0, wm, wn, half, j);
}

template <int NG, int TPG, int WM, int WN, int NTW0, int NTW1>
inline void launch_dma_variant(unsigned grid, size_t lds, hipStream_t stream, const ConvArgs& p) {
    static PerDeviceLatch raised;      // dynamic LDS beyond 64 KB: once per kernel and device
    // a failure here shows up as the launch error the caller reads with hipGetLastError()
    (void)raised.once([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_kernel<NG, TPG, WM, WN, NTW0, NTW1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    });
    hipLaunchKernelGGL((conv_dma_kernel<NG, TPG, WM, WN, NTW0, NTW1>), dim3(grid), dim3(64 * WM * WN), lds, stream,
                       p);
}

}  // namespace unetmi
