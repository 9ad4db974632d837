// One layer of the residual block, fused:
//     out = [src +] BN(PReLU(pointwise_1x1(depthwise_3x3_dilated(src))))        src = x [+ x2]
// (UNet_m.py:545-561: dw Conv2d(C,C,3x3,pad (1,d),dil (1,d),groups=C) -> Conv2d(C,C,1) -> PReLU -> BatchNorm2d,
//  layer 0 replaces the running value, layers 1..4 add to it), C = 64.
//
// The layer is bandwidth-bound (one read and one write of the activation is the floor), so the
// unfused form -- depthwise pass, pointwise pass, residual read: five tensor passes -- is replaced
// by one kernel that reads the tile (with its halo) once and writes once.  A workgroup owns a tile
// of 4 rows x 64 time steps, all 64 channels, and walks the channels in chunks of 8:
//   * the chunk's input patch [chunk][6][64+2d] is fetched with unconditional buffer loads (lanes
//     outside the image carry an out-of-range offset and read 0 = the zero padding) while the
//     previous chunk computes, then parked in LDS;
//   * the depthwise 3x3 runs on the vector ALUs (thread = one time step x 2 channels x 4 rows: the six
//     patch rows it reads serve all four output rows) and leaves its result in LDS as the B
//     operand [channel][pixel] of the pointwise product;
//   * the pointwise 64x64 product runs on v_mfma_f32_32x32x2_f32 with the whole weight matrix kept
//     in LDS in operand order for the life of the workgroup; wave w owns image row w of the tile.
// Workgroups are persistent over tiles so the weights are staged once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "../../include/unetmi.h"
#include "io_types.hpp"
#include "resblock_common.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DWISE = false drops the depthwise stage: a plain 64 -> 64 pointwise convolution (D = 0, no halo),
// whose input chunk is parked straight into the B-operand layout.
// TC = tile columns (64 or 32).  The narrower tile halves the accumulators (32 registers), which lets four
// workgroups share a compute unit instead of two: the layer is a chain of short phases (fetch, depthwise,
// MFMA, store) and more resident waves hide each other's latencies.
template <int D, bool DWISE, bool BF16, int TC>
__global__ __launch_bounds__(256, TC == 32 ? 4 : 2) void resblock_layer_kernel(const ResblockArgs a) {
    constexpr int kRbTC = TC;
    constexpr int NH = TC / 32;                           // 32-pixel MFMA tiles per wave (one image row of the tile)
    constexpr int kRbBPitch = kRbTR * TC + 32;            // B-operand row pitch: the two lane halves of an MFMA read
                                                          // (channels 2k, 2k+1) land in different banks
    constexpr int DG = 256 / TC;                          // depthwise thread groups (each takes CK/DG channels)
    constexpr int EB = BF16 ? 2 : 4;                      // bytes of a stored activation
    constexpr int kRbCK = DWISE ? 8 : 16;                 // channels per chunk
    constexpr int NCH = kRbC / kRbCK;
    constexpr int TCP = kRbTC + 2 * D;                    // patch columns
    constexpr int PROWS = kRbTR + (DWISE ? 2 : 0);        // patch rows
    constexpr int PLANE = PROWS * TCP;
    constexpr int NEL = kRbCK * PLANE;                    // patch elements of a chunk
    constexpr int NSLOT = (NEL + 255) / 256;
    constexpr int kOob = (int)0x80000000;
    extern __shared__ float smem[];
    float* s_in = smem;                                   // [NSLOT*256] patch of one chunk (DWISE only)
    float* s_b = smem + (DWISE ? NSLOT * 256 : 0);        // [chunk][kRbBPitch] B operand of the pointwise product
    float* s_w = s_b + kRbCK * kRbBPitch;                 // [32 k-steps][2 tiles][64 lanes] A operands
    float* s_par = s_w + kRbC * kRbC;                     // [3][64] pointwise bias, BatchNorm scale, shift
    float* s_dw = s_par + 3 * kRbC;                       // [64][10] depthwise taps + bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction; tell the compiler
    const int HW = a.H * a.W;

    // pointwise weights in MFMA operand order: A operand of k-step ks, output tile t, lane l is
    // W[t*32 + (l&31)][2*ks + (l>>5)]; written once, read conflict-free by every wave
    for (int i = tid; i < kRbC * kRbC; i += 256) {
        const int l = i & 63, t = (i >> 6) & 1, ks = i >> 7;
        s_w[i] = a.pw_w[(t * 32 + (l & 31)) * kRbC + 2 * ks + (l >> 5)];
    }
    if (tid < kRbC) {
        s_par[tid] = a.pw_b ? a.pw_b[tid] : 0.f;
        s_par[kRbC + tid] = a.scale[tid];
        s_par[2 * kRbC + tid] = a.shift[tid];
    }
    if (DWISE) {
        for (int i = tid; i < kRbC * 10; i += 256) {
            const int ch = i / 10, k = i - ch * 10;
            s_dw[i] = k < 9 ? a.dw_w[ch * 9 + k] : (a.dw_b ? a.dw_b[ch] : 0.f);
        }
    }
    const float alpha = a.alpha ? a.alpha[0] : 1.f;

    // depthwise work of this thread: column dcol of the tile, CK/DG of the chunk's channels
    const int dcol = tid % TC, dch0 = (tid / TC) * (kRbCK / DG);

    int voff[NSLOT];
    float stage[NSLOT];
    // patch element e = slot*256 + tid -> (channel, row, col) -> byte offset within chunk 0 of sample n
    auto aim = [&](int tile, int& n, int& row0, int& col0) {
        n = tile / (a.tiles_h * a.tiles_w);
        const int rem = tile - n * a.tiles_h * a.tiles_w;
        row0 = (rem / a.tiles_w) * kRbTR;
        col0 = (rem % a.tiles_w) * kRbTC;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int e = s * 256 + tid;
            const int ch = e / PLANE, r = (e % PLANE) / TCP, c = e % TCP;
            const int grow = row0 - (DWISE ? 1 : 0) + r, gcol = col0 - D + c;
            const bool ok = e < NEL && (unsigned)grow < (unsigned)a.H && (unsigned)gcol < (unsigned)a.W;
            voff[s] = ok ? ((ch * a.H + grow) * a.W + gcol) * EB : kOob;
        }
    };
    auto fetch = [&](int n, int chunk) {
        const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(a.x) + (size_t)n * kRbC * HW * EB), 0, kRbC * HW * EB, 0x00020000);
        const int soff = chunk * kRbCK * HW * EB;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) stage[s] = buf_ld_act<BF16>(rs0, voff[s], soff);
        if (a.x2) {
            const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(static_cast<const char*>(a.x2) + (size_t)n * kRbC * HW * EB), 0, kRbC * HW * EB, 0x00020000);
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) stage[s] += buf_ld_act<BF16>(rs1, voff[s], soff);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int e = s * 256 + tid;
            if (DWISE) s_in[e] = stage[s];
            else s_b[(e / PLANE) * kRbBPitch + (e % PLANE)] = stage[s];   // PLANE = tile pixels: [channel][pixel]
        }
    };

    int tile = blockIdx.x, n = 0, row0 = 0, col0 = 0;
    if (tile < a.ntiles) {
        aim(tile, n, row0, col0);
        fetch(n, 0);
    }
    while (tile < a.ntiles) {
        park();
        __syncthreads();

        f32x16 acc[2][NH];             // [output-channel tile][pixel tile of this wave's row]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][h][r] = 0.f;

        // residual values of this wave's outputs, fetched while the last chunk computes.  Output mapping
        // (channels are the MFMA rows, pixels its columns): lane l holds pixel h*32 + (l&31) of image row
        // `orow`, register r channel t*32 + (r&3) + 8*(r>>2) + 4*(l>>5); a store instruction writes two
        // contiguous 128-byte runs, which measured faster than 16 bytes per lane scattered over 64 channels.
        const int orow = row0 + wave;
        int obase[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int ocol = col0 + h * 32 + (lane & 31);
            obase[h] = (orow < a.H && ocol < a.W) ? ((4 * (lane >> 5)) * HW + orow * a.W + ocol) * EB : kOob;
        }
        float res[2][NH][16];
        auto fetch_residual = [&]() {
            const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(static_cast<const char*>(a.x) + (size_t)n * kRbC * HW * EB), 0,
                a.residual ? kRbC * HW * EB : 0, 0x00020000);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[t][h][r] = buf_ld_act<BF16>(rs_res, obase[h], (t * 32 + (r & 3) + 8 * (r >> 2)) * HW * EB);
        };

        // one chunk: start the next fetch (or, on the last chunk, the residual loads), depthwise stage,
        // pointwise MFMAs, park what was fetched.  The last chunk is peeled so that the residual
        // registers and the staging registers are never live together.
        auto do_chunk = [&](int chunk, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            if (!LAST) fetch(n, chunk + 1);
            else fetch_residual();
            if (DWISE) {
                // ---- depthwise 3x3, dilation (1, D) ---- (one channel at a time: keeps the live set small)
#pragma nounroll
                for (int cc = 0; cc < kRbCK / DG; ++cc) {
                    const int chl = dch0 + cc, chg = chunk * kRbCK + chl;
                    float k[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) k[i] = s_dw[chg * 10 + i];
                    const float b = s_dw[chg * 10 + 9];
                    float v[PROWS][3];
#pragma unroll
                    for (int r = 0; r < PROWS; ++r)
#pragma unroll
                        for (int q = 0; q < 3; ++q) v[r][q] = s_in[chl * PLANE + r * TCP + dcol + q * D];
#pragma unroll
                    for (int r = 0; r < kRbTR; ++r) {
                        float o = b;
#pragma unroll
                        for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                            for (int q = 0; q < 3; ++q) o = fmaf(v[r + kr][q], k[kr * 3 + q], o);
                        s_b[chl * kRbBPitch + r * kRbTC + dcol] = o;
                    }
                }
                __syncthreads();
            }
            // ---- pointwise product over this chunk's input channels ----
#pragma unroll
            for (int ks = 0; ks < kRbCK / 2; ++ks) {
                const int gks = chunk * (kRbCK / 2) + ks;
                float bv[NH];
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    bv[h] = s_b[(2 * ks + (lane >> 5)) * kRbBPitch + wave * kRbTC + h * 32 + (lane & 31)];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float av = s_w[(gks * 2 + t) * 64 + lane];
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        acc[t][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[h], acc[t][h], 0, 0, 0);
                }
            }
            if (!DWISE) __syncthreads();                  // everyone is done reading s_b before it is refilled
            if (!LAST) park();
            __syncthreads();
        };
#pragma nounroll
        for (int chunk = 0; chunk < NCH - 1; ++chunk) do_chunk(chunk, std::false_type{});
        do_chunk(NCH - 1, std::true_type{});

        const int cur_n = n;
        const int next = tile + gridDim.x;

        // ---- epilogue: bias, PReLU and BatchNorm affine in the layer's order, optional residual ----
        // buffer stores with 32-bit offsets; lanes outside the image carry an out-of-range offset
        const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<char*>(a.out) + (size_t)cur_n * kRbC * HW * EB, 0,
                                                              kRbC * HW * EB, 0x00020000);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cou = t * 32 + (r & 3) + 8 * (r >> 2);            // + 4*(lane>>5), folded into obase
                const int co = cou + 4 * (lane >> 5);
                const float pb = s_par[co], sc = s_par[kRbC + co], sh = s_par[2 * kRbC + co];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    float v = acc[t][h][r] + pb;
                    if (a.act_first) {
                        v = v >= 0.f ? v : alpha * v;
                        v = fmaf(v, sc, sh);
                    } else {
                        v = fmaf(v, sc, sh);
                        v = v >= 0.f ? v : alpha * v;
                    }
                    v += res[t][h][r];
                    buf_st_act<BF16>(v, rs_out, obase[h], cou * HW * EB);
                }
            }
        if (next < a.ntiles) {
            aim(next, n, row0, col0);
            fetch(n, 0);
        }
        tile = next;
    }
}

template <int D, bool DWISE, bool BF16, int TC>
static hipError_t launch_resblock_t(int grid, hipStream_t s, const ResblockArgs& a) {
    constexpr int kRbCK = DWISE ? 8 : 16;
    constexpr int NSLOT = (kRbCK * (kRbTR + (DWISE ? 2 : 0)) * (TC + 2 * D) + 255) / 256;
    constexpr size_t lds = (size_t)((DWISE ? NSLOT * 256 : 0) + kRbCK * (kRbTR * TC + 32) + kRbC * kRbC + 3 * kRbC + 10 * kRbC) * sizeof(float);
    static PerDeviceLatch raised;                   // dynamic LDS beyond 64 KB: once per kernel and device
    const hipError_t e = raised.once([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_layer_kernel<D, DWISE, BF16, TC>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((resblock_layer_kernel<D, DWISE, BF16, TC>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

// Tile width: 64 columns.  The 32-column instantiation (half the accumulators, four resident workgroups
// instead of two) measured slower inside ResUNet for both storage types; it stays selectable here.
static int tile_cols(bool) { return 64; }

template <int D, bool DWISE>
static hipError_t launch_resblock(int grid, hipStream_t s, const ResblockArgs& a, bool bf16) {
    return bf16 ? launch_resblock_t<D, DWISE, true, 64>(grid, s, a) : launch_resblock_t<D, DWISE, false, 64>(grid, s, a);
}

static int fill_tiles(ResblockArgs& a, bool bf16) {
    const int tc = tile_cols(bf16);
    a.tiles_h = (a.H + kRbTR - 1) / kRbTR; a.tiles_w = (a.W + tc - 1) / tc;
    const long long nt = (long long)a.N * a.tiles_h * a.tiles_w;
    if (nt > 0x7fffffffll) return -1;
    a.ntiles = (int)nt;
    return (int)std::min<long long>(nt, 256 * 4 * 4);       // persistent: a few tiles per workgroup
}

}  // namespace unetmi

using namespace unetmi;

extern "C" {

int unetmi_resblock_layer_fwd(int N, int C, int H, int W, int dilation_w, const void* x, const void* x2,
                              const float* dw_weight, const float* dw_bias, const float* pw_weight,
                              const float* pw_bias, const float* prelu_alpha, const float* bn_scale,
                              const float* bn_shift, int residual, void* out, int dtype, void* stream) {
    if (N < 0 || H < 1 || W < 1) return api_fail(UNETMI_E_BAD_DESC, "resblock_layer: bad sizes");
    if (C != kRbC) return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: built for 64 channels");
    if ((long long)C * H * W * 4 >= (1ll << 31)) return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: sample larger than 2 GiB");
    if (N == 0) return UNETMI_OK;
    if (!x || !dw_weight || !pw_weight || !prelu_alpha || !bn_scale || !bn_shift || !out)
        return api_fail(UNETMI_E_BAD_ARG, "resblock_layer: null pointer");
    if (out == x || (x2 && out == x2)) return api_fail(UNETMI_E_BAD_ARG, "resblock_layer: out must not alias an input");
    if (x2 && residual) return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: an input addend together with the residual add is not built");
    ResblockArgs a{};
    a.x = x; a.x2 = x2; a.dw_w = dw_weight; a.dw_b = dw_bias; a.pw_w = pw_weight; a.pw_b = pw_bias;
    a.alpha = prelu_alpha; a.scale = bn_scale; a.shift = bn_shift; a.out = out;
    a.N = N; a.H = H; a.W = W; a.residual = residual ? 1 : 0; a.act_first = 1;
    const bool bf16 = dtype == UNETMI_DTYPE_BF16;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dilation_w != 1 && dilation_w != 2 && dilation_w != 4 && dilation_w != 8 && dilation_w != 16)
        return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: time dilation must be 1, 2, 4, 8 or 16");
    // one input: the LDS-DMA pipeline; with an input addend: the register-staged kernel.  The 16-byte DMA groups need
    // 4-byte-aligned addresses: always true for fp32, for bfloat16 only when every row starts on an even element.
    if (!x2 && (long long)C * H * W * 4 < 0x20000000ll && H < 30000 && (!bf16 || (W & 1) == 0)) {
        const hipError_t e = launch_resblock_dma(dilation_w, s, a, bf16);
        return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("resblock_layer launch: ") + hipGetErrorString(e));
    }
    const int grid = fill_tiles(a, bf16);
    if (grid < 0) return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: too many tiles");
    hipError_t e;
    switch (dilation_w) {
        case 1: e = launch_resblock<1, true>(grid, s, a, bf16); break;
        case 2: e = launch_resblock<2, true>(grid, s, a, bf16); break;
        case 4: e = launch_resblock<4, true>(grid, s, a, bf16); break;
        case 8: e = launch_resblock<8, true>(grid, s, a, bf16); break;
        case 16: e = launch_resblock<16, true>(grid, s, a, bf16); break;
        default: return api_fail(UNETMI_E_UNSUPPORTED, "resblock_layer: time dilation must be 1, 2, 4, 8 or 16");
    }
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("resblock_layer launch: ") + hipGetErrorString(e));
}

int unetmi_pwconv64_fwd(int N, int H, int W, const void* x, const void* x2, const float* weight, const float* bias,
                        const float* scale, const float* shift, const float* leaky_slope, int act_first, int residual,
                        void* out, int dtype, void* stream) {
    if (N < 0 || H < 1 || W < 1) return api_fail(UNETMI_E_BAD_DESC, "pwconv64: bad sizes");
    if ((long long)kRbC * H * W * 4 >= (1ll << 31)) return api_fail(UNETMI_E_UNSUPPORTED, "pwconv64: sample larger than 2 GiB");
    if (N == 0) return UNETMI_OK;
    if (!x || !weight || !scale || !shift || !out) return api_fail(UNETMI_E_BAD_ARG, "pwconv64: null pointer");
    if (out == x || (x2 && out == x2)) return api_fail(UNETMI_E_BAD_ARG, "pwconv64: out must not alias an input");
    if (x2 && residual) return api_fail(UNETMI_E_UNSUPPORTED, "pwconv64: an input addend together with the residual add is not built");
    ResblockArgs a{};
    a.x = x; a.x2 = x2; a.pw_w = weight; a.pw_b = bias; a.alpha = leaky_slope; a.scale = scale; a.shift = shift;
    a.out = out; a.N = N; a.H = H; a.W = W; a.residual = residual ? 1 : 0; a.act_first = act_first ? 1 : 0;
    const int grid = fill_tiles(a, dtype == UNETMI_DTYPE_BF16);
    if (grid < 0) return api_fail(UNETMI_E_UNSUPPORTED, "pwconv64: too many tiles");
    const hipError_t e = launch_resblock<0, false>(grid, static_cast<hipStream_t>(stream), a, dtype == UNETMI_DTYPE_BF16);
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("pwconv64 launch: ") + hipGetErrorString(e));
}

}  // extern "C"
