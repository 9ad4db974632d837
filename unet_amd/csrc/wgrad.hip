// Weight gradient of Conv2d / ConvTranspose2d on the fp32 matrix cores.
//
// Both cases are one correlation
//     dW[a][b][r][s] = sum_{n,p,q} P[n,a,p,q] * Q[n,b, p*sh - ph + r*dh, q*sw - pw + s*dw]
// Conv2d:          P = gradient of the conv output [N,K,OH,OW], Q = layer input [N,C,H,W]  -> dW[K][C][kh][kw]
// ConvTranspose2d: P = layer input [N,Cin,H,W] (optionally two tensors concatenated along channels),
//                  Q = gradient of the output [N,Cout,OH,OW]                                 -> dW[Cin][Cout][kh][kw]
// which are torch's weight layouts, so one kernel serves both.
//
// The reduction runs over pixels, so it maps to v_mfma_f32_16x16x4_f32 with k = 4 consecutive
// pixels of one row: A operand = P[a = lane&15][pixel lane>>4], B operand = Q[b = lane&15][shifted
// pixel lane>>4], D = a 16x16 block of dW for one tap.  A workgroup owns 64 a-channels x (BT
// b-channels x all taps); the (b-tile, tap) pairs are dealt round-robin to its four waves (up to 9
// each = 144 accumulator registers).  It walks a 32-pixel-wide column strip of one sample top to
// bottom: the P row tile is double-buffered in LDS, the Q rows live in a ring of (kh-1)*dh+1+sh rows
// so every input row is fetched once per strip, and the global loads of step p+1 are in flight in
// registers while step p's MFMAs run.  Partial results of the (few hundred) workgroups that share an
// output block go to a workspace [split][tap][a][b] and a second kernel adds them in a fixed order:
// no atomics, bit-reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "../../include/unetmi.h"
#include "io_types.hpp"

namespace unetmi {

int api_fail(int code, const std::string& msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTQ = 32;          // pixels of a P row handled per step
constexpr int kPPitch = 34;      // LDS pitch of a P row (2 mod 4: lanes (a, q) of a 32-lane half read banks 2a + q)
constexpr int kMaxQ = 12;        // Q elements staged per thread per step
constexpr int kMaxPairs = 36;    // (b-tile, tap) pairs per workgroup = 4 waves x 9

struct WgradArgs {
    const float* p0;
    const float* p1;
    const float* q;
    float* ws;
    int N, A0, A1, A, B, PH, PW, QH, QW, kh, kw, sh, sw, ph, pw, dh, dw;
    int BT;          // b-channels per workgroup (multiple of 16)
    int qcols;       // staged columns of a Q row: (kTQ-1)*sw + (kw-1)*dw + 1
    int qpitch;      // LDS pitch of a Q row (2 mod 4)
    int R;           // rows of the ring
    int win;         // rows a step reads: (kh-1)*dh + 1
    int nqc;         // column strips per sample
    int nrs, rps;    // row segments per column strip, P rows per segment
    int strips, spw; // N*nqc*nrs, strips per workgroup
    int ntaps, npairs;
    int nq_items;    // sh*BT*qcols elements staged per step
    int ring_bias;   // multiple of R added before the modulo so negative rows map consistently
};

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

constexpr int kOob = (int)0x80000000;     // byte offset beyond any buffer: the load returns 0

#ifndef WG_ABLATE
#define WG_ABLATE 0     // timing builds only (tools/wgrad_probe.py): 1 no global loads, 2 no LDS stores in the loop, 4 no MFMAs, 8 no step barrier
#endif

template <int NT>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs g) {
    extern __shared__ float lds[];
    float* sP = lds;                               // [2][64][kPPitch]
    float* sQ = lds + 2 * 64 * kPPitch;            // [R][BT][qpitch], then 256 floats nobody reads
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rowstride = g.BT * g.qpitch;
    const int bb = blockIdx.y * g.BT;
    // a-blocks enumerate 64-channel blocks of the first source, then of the second
    const int nblk0 = (g.A0 + 63) >> 6;
    const bool second = (int)blockIdx.z >= nblk0;
    const int a_lo = (second ? (int)blockIdx.z - nblk0 : (int)blockIdx.z) * 64;   // first channel within its source
    const int a_cnt = second ? g.A1 : g.A0;                                       // channels of that source
    const int a_out = a_lo + (second ? g.A0 : 0);                                 // first row of dW
    const float* psrc = second ? g.p1 : g.p0;

    // this wave's (b-tile, tap) pairs
    int pair_row[NT], pair_off[NT], pair_tap[NT], pair_bt[NT];
    bool pair_ok[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int idx = wave + 4 * j;
        pair_ok[j] = idx < g.npairs;
        const int ic = pair_ok[j] ? idx : 0;
        const int bt = ic / g.ntaps, tap = ic - bt * g.ntaps;
        const int r = tap / g.kw, s = tap - r * g.kw;
        pair_row[j] = r * g.dh;
        pair_off[j] = s * g.dw + bt * 16 * g.qpitch;
        pair_tap[j] = tap;
        pair_bt[j] = bt;
    }
    const int la = (lane & 15) * kPPitch + (lane >> 4);
    const int lb = (lane & 15) * g.qpitch + (lane >> 4) * g.sw;

    f32x4 acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int pq = tid & 31, pa = tid >> 5;                 // P element i: channel a_lo + pa + 8*i, pixel pq
    const int dummy = g.R * rowstride + tid;

    int buf = 0;
    const int st_lo = blockIdx.x * g.spw, st_hi = min(st_lo + g.spw, g.strips);
    for (int st = st_lo; st < st_hi; ++st) {
        const int cs = st / g.nrs, seg = st - cs * g.nrs;          // column strip, row segment within it
        const int n = cs / g.nqc, q0 = (cs - n * g.nqc) * kTQ;
        const int p_lo = seg * g.rps, p_hi = min(p_lo + g.rps, g.PH);
        const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(psrc + (size_t)n * a_cnt * g.PH * g.PW), 0, a_cnt * g.PH * g.PW * 4, 0x00020000);
        const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.q + (size_t)n * g.B * g.QH * g.QW), 0, g.B * g.QH * g.QW * 4, 0x00020000);

        int p_voff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = a_lo + pa + 8 * i;
            p_voff[i] = (a < a_cnt && q0 + pq < g.PW) ? (a * g.PH * g.PW + q0 + pq) * 4 : kOob;
        }
        // Q element i of this thread: flat index tid + 256*i over [sh rows][BT channels][qcols]
        int q_voff[kMaxQ], q_lds[kMaxQ], q_rs[kMaxQ];
#pragma unroll
        for (int i = 0; i < kMaxQ; ++i) {
            const int idx = tid + 256 * i;
            const int per_row = g.BT * g.qcols;
            const int rs = idx / per_row, rem = idx - rs * per_row;
            const int bl = rem / g.qcols, cl = rem - bl * g.qcols;
            const int b = bb + bl, col = q0 * g.sw - g.pw + cl;
            const bool have = idx < g.nq_items;
            q_rs[i] = have ? rs : -1;
            q_lds[i] = bl * g.qpitch + cl;
            q_voff[i] = (have && b < g.B && col >= 0 && col < g.QW) ? (b * g.QH * g.QW + col) * 4 : kOob;
        }

        float rp[8], rq[kMaxQ];
        auto load_p = [&](int p) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rp[i] = (WG_ABLATE & 1) ? (float)p : buf_load(rs_p, p_voff[i], p * g.PW * 4);
        };
        auto store_p = [&](int b2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) sP[(b2 * 64 + pa + 8 * i) * kPPitch + pq] = rp[i];
        };
        // rows y0 .. y0+sh-1 of Q
        auto load_q = [&](int y0) {
#pragma unroll
            for (int i = 0; i < kMaxQ; ++i) {
                const int y = y0 + q_rs[i];
                rq[i] = (WG_ABLATE & 1) ? (float)y : buf_load(rs_q, ((unsigned)y < (unsigned)g.QH && q_rs[i] >= 0) ? q_voff[i] + y * g.QW * 4 : kOob, 0);
            }
        };
        auto store_q = [&](int y0) {
            const int s0 = (y0 + g.ring_bias) % g.R;             // wave-uniform
#pragma unroll
            for (int i = 0; i < kMaxQ; ++i) {
                int slot = s0 + q_rs[i];
                slot = slot >= g.R ? slot - g.R : slot;
                sQ[q_rs[i] >= 0 ? slot * rowstride + q_lds[i] : dummy] = rq[i];
            }
        };

        // prologue: the window of the segment's first step and its P tile
        const int base0 = p_lo * g.sh - g.ph;
        for (int y = base0; y < base0 + g.win; y += g.sh) {
            load_q(y);
            store_q(y);
        }
        load_p(p_lo);
        store_p(buf);
        __syncthreads();

        int rowoff[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) rowoff[j] = ((base0 + pair_row[j] + g.ring_bias) % g.R) * rowstride + pair_off[j] + lb;

        for (int p = p_lo; p < p_hi; ++p) {
            const int ynew = p * g.sh - g.ph + g.win;        // first row step p+1 needs beyond this window
            const bool more = p + 1 < p_hi;
            if (more) {
                load_p(p + 1);
                load_q(ynew);
            }
            // Operands of a 4-pixel group are read one group ahead into a second register set: this kernel runs one
            // wave per SIMD, so a read that is waited for right where it is issued (what the compiler makes of the
            // plain loop: one lgkmcnt(0) per four MFMAs) leaves the matrix pipe idle for the whole LDS round trip.
            const float* tp = sP + buf * 64 * kPPitch + la;
            float av[2][4], bv[2][NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[0][t] = tp[t * 16 * kPPitch];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[0][j] = sQ[rowoff[j]];
#pragma unroll
            for (int gq = 0; gq < kTQ / 4; ++gq) {
                const int cur = gq & 1, nxt = cur ^ 1;
                if (gq + 1 < kTQ / 4) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[nxt][t] = tp[t * 16 * kPPitch + (gq + 1) * 4];
#pragma unroll
                    for (int j = 0; j < NT; ++j) bv[nxt][j] = sQ[rowoff[j] + (gq + 1) * 4 * g.sw];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (!(WG_ABLATE & 4)) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][t], bv[cur][j], acc[j][t], 0, 0, 0);
                        else acc[j][t][0] += av[cur][t] * bv[cur][j];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more && !(WG_ABLATE & 2)) {
                store_p(buf ^ 1);
                store_q(ynew);
            }
            if (!(WG_ABLATE & 8)) __syncthreads();
            buf ^= 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                rowoff[j] += g.sh * rowstride;
                // the slot index is rowoff's multiple of rowstride; wrap when it passes the ring end
                if (rowoff[j] - pair_off[j] - lb >= g.R * rowstride) rowoff[j] -= g.R * rowstride;
            }
        }
    }

    // partial block of dW -> ws[split][tap][a][b]
    float* dst = g.ws + (size_t)blockIdx.x * g.ntaps * g.A * g.B;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (!pair_ok[j]) continue;
        const int b = bb + pair_bt[j] * 16 + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int al = a_lo + t * 16 + 4 * (lane >> 4) + i;
                if (al < a_cnt && b < g.B)
                    dst[((size_t)pair_tap[j] * g.A + a_out - a_lo + al) * g.B + b] = acc[j][t][i];
            }
    }
}

// dW[a][b][tap] = sum over splits of ws[split][tap][a][b], fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits,
                                                           int ntaps, int A, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = ntaps * A * B;
    if (i >= total) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * total + i];
    const int tap = i / (A * B), ab = i - tap * (A * B);
    dw[(size_t)ab * ntaps + tap] = s;
}

// The same sum when there are many splits and few outputs (the thin and the deep layers): one thread per output
// walks `splits` dependent-latency loads, ~100 us for a thousand of them.  Here 16 lanes share an output -- lane q
// sums splits q, q+16, ... (four loads in flight), and the 16 partial sums are added in index order -- so the
// result is still a fixed-order sum, just a different fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                int splits, int ntaps, int A, int B) {
    __shared__ float part[16][17];
    const int il = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int total = ntaps * A * B;
    const int i = blockIdx.x * 16 + il;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int k = q;
        for (; k + 48 < splits; k += 64) {
            s0 += ws[(size_t)k * total + i];
            s1 += ws[(size_t)(k + 16) * total + i];
            s2 += ws[(size_t)(k + 32) * total + i];
            s3 += ws[(size_t)(k + 48) * total + i];
        }
        for (; k < splits; k += 16) s0 += ws[(size_t)k * total + i];
    }
    part[q][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && i < total) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += part[r][il];
        const int tap = i / (A * B), ab = i - tap * (A * B);
        dw[(size_t)ab * ntaps + tap] = s;
    }
}


// ---- thin layers ------------------------------------------------------------------------------------------------
// Weight gradient when the SHIFTED operand Q has one or two channels and the kernel is one column wide with unit stride
// and dilation (the UNet's last decoder 96 -> 1, its first encoder 2 -> 32, the 1 -> 1 head): ~1 GFLOP over hundreds
// of megabytes, i.e. a streaming reduction, which the matrix kernel above serves at a fifth of the memory rate.
// A workgroup owns channel a of P, a 64-column strip and a chunk of rows, and walks every sample: a thread (lane =
// column) loads 8 consecutive rows of P and the 8 + KH - 1 rows of each Q channel they meet (Q is a few megabytes and
// stays in L2 while every channel of P streams past once), and keeps BB x KH sums.  One partial per workgroup goes to
// ws[split][tap][a][b]; the fixed-order reduction kernels above finish the sum.
struct ThinWgradArgs {
    const float* p0; const float* p1; const float* q; float* ws;
    int N, A0, A1, PH, PW, QH, ph, chunk_rows;
};

template <int KH, int BB>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const ThinWgradArgs g) {
    constexpr int RPT = 8, NQ = RPT + KH - 1;
    __shared__ float s_red[4][BB * KH];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // one XCD walks a contiguous run of (channel, row chunk, strip) tiles -- strips fastest -- so the 128-byte lines that
    // neighbouring strips share (rows are not line-aligned) are fetched by one L2
    const unsigned tiles = gridDim.x * gridDim.y * gridDim.z;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = lin & 7u, q8 = tiles >> 3, r8 = tiles & 7u;
    const unsigned t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
    const int strip = (int)(t % gridDim.x), chunk = (int)((t / gridDim.x) % gridDim.y), a = (int)(t / (gridDim.x * gridDim.y));
    const int col = strip * 64 + lane;
    const bool col_ok = col < g.PW;
    const bool second = a >= g.A0;
    const float* pa = second ? g.p1 + (size_t)(a - g.A0) * g.PH * g.PW : g.p0 + (size_t)a * g.PH * g.PW;
    const size_t p_sample = (size_t)(second ? g.A1 : g.A0) * g.PH * g.PW;
    const size_t q_plane = (size_t)g.QH * g.PW;
    const int r_lo = chunk * g.chunk_rows, r_hi = min(r_lo + g.chunk_rows, g.PH);

    float acc[BB][KH];
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int r = 0; r < KH; ++r) acc[b][r] = 0.f;

    for (int n = 0; n < g.N; ++n) {
        const float* pn = pa + (size_t)n * p_sample;
        const float* qn = g.q + (size_t)n * BB * q_plane;
        for (int p0 = r_lo + wave * RPT; p0 < r_hi; p0 += 4 * RPT) {
            float pv[RPT], qv[BB][NQ];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const bool ok = col_ok && p0 + i < r_hi;
                pv[i] = ok ? pn[(size_t)(p0 + i) * g.PW + col] : 0.f;
            }
#pragma unroll
            for (int b = 0; b < BB; ++b)
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int y = p0 - g.ph + j;
                    const bool ok = col_ok && (unsigned)y < (unsigned)g.QH;
                    qv[b][j] = ok ? qn[(size_t)b * q_plane + (size_t)y * g.PW + col] : 0.f;
                }
#pragma unroll
            for (int b = 0; b < BB; ++b)
#pragma unroll
                for (int r = 0; r < KH; ++r)
#pragma unroll
                    for (int i = 0; i < RPT; ++i) acc[b][r] = fmaf(pv[i], qv[b][i + r], acc[b][r]);
        }
    }
    // sum over the 64 columns of a wave, then over the 4 waves, in a fixed order
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            float v = acc[b][r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) s_red[wave][b * KH + r] = v;
        }
    __syncthreads();
    if (threadIdx.x < BB * KH) {
        const int b = threadIdx.x / KH, r = threadIdx.x % KH;
        const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) + (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
        const int A = g.A0 + g.A1, split = chunk * gridDim.x + strip, total = KH * A * BB;
        g.ws[(size_t)split * total + ((size_t)r * A + a) * BB + b] = v;
    }
}

struct WgradPlan {
    WgradArgs a;
    int NT, splits, bgroups, ablocks;
    size_t lds_bytes, ws_bytes;
};

struct ThinWgradPlan { bool ok; int strips, chunks, chunk_rows, splits; size_t ws_bytes; };

// the thin kernel serves: Q of 1 or 2 channels, kernel 1, 3 or 7 rows x 1 column, unit stride and dilation, no column padding
static ThinWgradPlan plan_wgrad_thin(const WgradArgs& a) {
    ThinWgradPlan t{};
    t.ok = a.B <= 2 && a.kw == 1 && a.sw == 1 && a.sh == 1 && a.dw == 1 && a.dh == 1 && a.pw == 0 && a.PW == a.QW &&
           (a.kh == 1 || a.kh == 3 || a.kh == 7);
    if (!t.ok) return t;
    t.strips = (a.PW + 63) / 64;
    // about a thousand workgroups; a chunk is a whole number of 32-row passes
    int want = 1024 / std::max(1, t.strips * a.A);
    want = std::min(std::max(want, 1), 8);
    t.chunk_rows = std::max(32, ((a.PH + want - 1) / want + 31) / 32 * 32);
    t.chunks = (a.PH + t.chunk_rows - 1) / t.chunk_rows;
    t.splits = t.strips * t.chunks;
    t.ws_bytes = (size_t)t.splits * a.ntaps * a.A * a.B * sizeof(float);
    return t;
}

template <int KH, int BB>
static hipError_t launch_wgrad_thin(const WgradArgs& a, const ThinWgradPlan& t, hipStream_t s) {
    ThinWgradArgs g;
    g.p0 = a.p0; g.p1 = a.p1; g.q = a.q; g.ws = a.ws;
    g.N = a.N; g.A0 = a.A0; g.A1 = a.A1; g.PH = a.PH; g.PW = a.PW; g.QH = a.QH; g.ph = a.ph; g.chunk_rows = t.chunk_rows;
    hipLaunchKernelGGL((wgrad_thin_kernel<KH, BB>), dim3(t.strips, t.chunks, a.A), dim3(256), 0, s, g);
    return hipGetLastError();
}

static int round_pitch(int cols) {          // smallest pitch >= cols that is 2 mod 4 (see kPPitch)
    int p = cols;
    while ((p & 3) != 2) ++p;
    return p;
}

static int plan_wgrad(int N, int A0, int A1, int B, int PH, int PW, int QH, int QW, int kh, int kw, int sh, int sw,
                      int ph, int pw, int dh, int dw, WgradPlan* pl, std::string* why) {
    WgradArgs& a = pl->a;
    a = WgradArgs{};
    a.N = N; a.A0 = A0; a.A1 = A1; a.A = A0 + A1; a.B = B; a.PH = PH; a.PW = PW; a.QH = QH; a.QW = QW;
    a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw; a.dh = dh; a.dw = dw;
    a.ntaps = kh * kw;
    if ((long long)std::max(A0, A1) * PH * PW >= (1ll << 29) || (long long)B * QH * QW >= (1ll << 29)) {
        *why = "one sample of an operand exceeds 2 GiB"; return UNETMI_E_UNSUPPORTED;
    }
    if (a.ntaps > kMaxPairs) { *why = "more than 36 kernel taps"; return UNETMI_E_UNSUPPORTED; }
    a.qcols = (kTQ - 1) * sw + (kw - 1) * dw + 1;
    a.qpitch = round_pitch(a.qcols);
    a.win = (kh - 1) * dh + 1;
    a.R = a.win + sh;
    a.ring_bias = ((ph + a.R - 1) / a.R + 1) * a.R;
    const int btiles = (B + 15) / 16;
    int nbt = std::min(std::min(kMaxPairs / a.ntaps, btiles), 4);
    while (nbt > 1 && sh * nbt * 16 * a.qcols > kMaxQ * 256) --nbt;
    if (sh * nbt * 16 * a.qcols > kMaxQ * 256) { *why = "a row of the shifted operand does not fit the staging registers"; return UNETMI_E_UNSUPPORTED; }
    a.BT = nbt * 16;
    a.npairs = nbt * a.ntaps;
    a.nq_items = sh * a.BT * a.qcols;
    pl->NT = (a.npairs + 3) / 4;
    pl->bgroups = (B + a.BT - 1) / a.BT;
    pl->ablocks = (A0 + 63) / 64 + (A1 + 63) / 64;
    a.nqc = (PW + kTQ - 1) / kTQ;
    const int G = pl->bgroups * pl->ablocks;
    // a small batch of wide layers has few column strips: cut each into row segments (>= 16 rows, each pays one
    // window of prologue rows) until the grid holds two workgroups per CU
    a.nrs = 1;
    while (N * a.nqc * a.nrs * G < 512 && PH / (a.nrs + 1) >= 16) ++a.nrs;
    a.rps = (PH + a.nrs - 1) / a.nrs;
    a.nrs = (PH + a.rps - 1) / a.rps;
    a.strips = N * a.nqc * a.nrs;
    int splits = std::min(a.strips, std::max(1, (768 + G - 1) / G));
    a.spw = (a.strips + splits - 1) / splits;
    pl->splits = (a.strips + a.spw - 1) / a.spw;
    pl->lds_bytes = (size_t)(2 * 64 * kPPitch + a.R * a.BT * a.qpitch + 256) * sizeof(float);
    if (pl->lds_bytes > 160 * 1024) { *why = "ring of input rows exceeds LDS"; return UNETMI_E_UNSUPPORTED; }
    pl->ws_bytes = (size_t)pl->splits * a.ntaps * a.A * a.B * sizeof(float);
    const ThinWgradPlan thin = plan_wgrad_thin(a);           // thin layers take the streaming kernel and its workspace
    if (thin.ok) pl->ws_bytes = std::max(pl->ws_bytes, thin.ws_bytes);
    return UNETMI_OK;
}

template <int NT>
static hipError_t launch_wgrad(const WgradPlan& pl, hipStream_t s) {
    static PerDeviceLatch raised;                   // dynamic LDS beyond 64 KB: once per kernel and device
    const hipError_t e = raised.once([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<NT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((wgrad_kernel<NT>), dim3(pl.splits, pl.bgroups, pl.ablocks), dim3(256), pl.lds_bytes, s, pl.a);
    return hipGetLastError();
}

}  // namespace unetmi

using namespace unetmi;

extern "C" {

static int wgrad_plan_from_desc(const UnetmiConvDesc* d, WgradPlan* pl, const char* what) {
    if (!d) return api_fail(UNETMI_E_BAD_ARG, std::string(what) + ": null descriptor");
    if (d->N < 0 || d->C0 < 1 || d->C1 < 0 || d->K < 1 || d->H < 1 || d->W < 1 || d->OH < 1 || d->OW < 1 || d->kh < 1 ||
        d->kw < 1 || d->sh < 1 || d->sw < 1 || d->dh < 1 || d->dw < 1 || d->ph < 0 || d->pw < 0)
        return api_fail(UNETMI_E_BAD_DESC, std::string(what) + ": bad geometry");
    if (d->in_mode == UNETMI_IN_ADD) return api_fail(UNETMI_E_UNSUPPORTED, std::string(what) + ": summed inputs are not built");
    if (!d->transposed && d->C1 != 0) return api_fail(UNETMI_E_UNSUPPORTED, std::string(what) + ": concatenated input of a plain conv is not built");
    std::string why;
    const int N = std::max(d->N, 1);
    const int rc = d->transposed
        ? plan_wgrad(N, d->C0, d->C1, d->K, d->H, d->W, d->OH, d->OW, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, d->dh, d->dw, pl, &why)
        : plan_wgrad(N, d->K, 0, d->C0, d->OH, d->OW, d->H, d->W, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, d->dh, d->dw, pl, &why);
    if (rc) return api_fail(rc, std::string(what) + ": " + why);
    return UNETMI_OK;
}

size_t unetmi_conv2d_wgrad_workspace_bytes(const UnetmiConvDesc* d) {
    WgradPlan pl;
    if (wgrad_plan_from_desc(d, &pl, "conv2d_wgrad_workspace_bytes")) return 0;
    return pl.ws_bytes;
}

int unetmi_conv2d_wgrad(const UnetmiConvDesc* d, const float* in0, const float* in1, const float* dout, float* dweight,
                        void* ws, size_t ws_bytes, void* stream) {
    WgradPlan pl;
    if (int rc = wgrad_plan_from_desc(d, &pl, "conv2d_wgrad")) return rc;
    if (!dweight) return api_fail(UNETMI_E_BAD_ARG, "conv2d_wgrad: null gradient pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t wfloats = (size_t)(d->C0 + d->C1) * d->K * d->kh * d->kw;
    if (d->N == 0) {
        const hipError_t e = hipMemsetAsync(dweight, 0, wfloats * sizeof(float), s);
        return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("conv2d_wgrad memset: ") + hipGetErrorString(e));
    }
    if (!in0 || !dout || !ws || (d->C1 > 0 && !in1)) return api_fail(UNETMI_E_BAD_ARG, "conv2d_wgrad: null tensor pointer");
    if (ws_bytes < pl.ws_bytes) return api_fail(UNETMI_E_BAD_ARG, "conv2d_wgrad: workspace too small");
    if (d->transposed) { pl.a.p0 = in0; pl.a.p1 = d->C1 > 0 ? in1 : in0; pl.a.q = dout; }
    else               { pl.a.p0 = dout; pl.a.p1 = dout; pl.a.q = in0; }
    pl.a.ws = static_cast<float*>(ws);
    hipError_t e;
    const ThinWgradPlan thin = plan_wgrad_thin(pl.a);
    if (thin.ok) {
        const int kh = pl.a.kh;
#define UNETMI_THIN(KH_) (pl.a.B == 1 ? launch_wgrad_thin<KH_, 1>(pl.a, thin, s) : launch_wgrad_thin<KH_, 2>(pl.a, thin, s))
        e = kh == 1 ? UNETMI_THIN(1) : (kh == 3 ? UNETMI_THIN(3) : UNETMI_THIN(7));
#undef UNETMI_THIN
        if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("conv2d_wgrad (thin) launch: ") + hipGetErrorString(e));
        pl.splits = thin.splits;
    } else {
    switch (pl.NT) {
        case 1: e = launch_wgrad<1>(pl, s); break;
        case 2: e = launch_wgrad<2>(pl, s); break;
        case 3: e = launch_wgrad<3>(pl, s); break;
        case 4: e = launch_wgrad<4>(pl, s); break;
        case 5: e = launch_wgrad<5>(pl, s); break;
        case 6: e = launch_wgrad<6>(pl, s); break;
        case 7: e = launch_wgrad<7>(pl, s); break;
        case 8: e = launch_wgrad<8>(pl, s); break;
        default: e = launch_wgrad<9>(pl, s); break;
    }
    if (e != hipSuccess) return api_fail(UNETMI_E_HIP, std::string("conv2d_wgrad launch: ") + hipGetErrorString(e));
    }
    const int total = pl.a.ntaps * pl.a.A * pl.a.B;
    // one thread per output is a chain of `splits` load latencies: with many splits, 16 lanes share an output
    if (pl.splits >= 32)
        hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((total + 15) / 16), dim3(256), 0, s, pl.a.ws, dweight, pl.splits,
                           pl.a.ntaps, pl.a.A, pl.a.B);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, pl.a.ws, dweight, pl.splits,
                           pl.a.ntaps, pl.a.A, pl.a.B);
    e = hipGetLastError();
    return e == hipSuccess ? UNETMI_OK : api_fail(UNETMI_E_HIP, std::string("conv2d_wgrad reduce launch: ") + hipGetErrorString(e));
}

}  // extern "C"
