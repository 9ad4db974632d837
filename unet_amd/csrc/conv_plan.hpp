// Host-side plan of the implicit-GEMM convolution: how a (transposed)
// convolution is cut into output phases, taps, channel chunks and workgroup
// tiles.  Shared by the weight packer and the launcher so that the packed
// image and the kernel's read order cannot drift apart.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/unetmi.h"

namespace unetmi {

constexpr int kMaxPhase = 4;      // output phases per axis = stride of a transposed conv
constexpr int kWaves = 4;         // waves per workgroup
constexpr int kTileJ = 32;        // output columns of one MFMA tile (the 32 of 32x32x2)
constexpr int kLdsBudget3 = 25 * 1024;  // one staged (input patch + weights) image such that three
                                       // double-buffered workgroups share a CU's 160 KB ...
constexpr int kLdsBudget = 38 * 1024;   // ... or two, when a channel pair needs more (or 128 accumulators cap it at two)
constexpr int kMaxInSlots = 24;   // input-patch floats a thread prefetches per chunk (x256 threads)
constexpr int kMaxW4Slots = 10;   // weight float4s a thread prefetches per chunk
constexpr int kMaxTaps = 64;      // taps of one (row phase, column phase) pair

// One spatial axis.  Output index o = nphase * q + phase (q = "phase index").
// Tap t of that phase reads input index  q * in_step + base[phase] + t * tap_step
// and weight index  kidx0[phase] + t * kidx_step[phase].
struct AxisPlan {
    int nphase = 1;
    int in_step = 1;
    int tap_step = 1;
    int ntap[kMaxPhase] = {0, 0, 0, 0};
    int base[kMaxPhase] = {0, 0, 0, 0};
    int kidx0[kMaxPhase] = {0, 0, 0, 0};
    int kidx_step[kMaxPhase] = {1, 1, 1, 1};
    int nout[kMaxPhase] = {0, 0, 0, 0};   // phase indices that land inside the output
};

inline bool plan_axis(bool transposed, int in, int out, int k, int s, int p, int d, int op,
                      AxisPlan* ax, std::string* err) {
    if (k < 1 || s < 1 || d < 1 || p < 0 || in < 1) { *err = "non-positive geometry"; return false; }
    if (!transposed) {
        const int expect = (in + 2 * p - d * (k - 1) - 1) / s + 1;
        if (in + 2 * p - d * (k - 1) - 1 < 0 || out != expect) {
            *err = "conv output size " + std::to_string(out) + " != " + std::to_string(expect);
            return false;
        }
        ax->nphase = 1; ax->in_step = s; ax->tap_step = d;
        ax->ntap[0] = k; ax->base[0] = -p; ax->kidx0[0] = 0; ax->kidx_step[0] = 1; ax->nout[0] = out;
        return true;
    }
    const int expect = (in - 1) * s - 2 * p + d * (k - 1) + op + 1;
    if (out != expect || out < 1) {
        *err = "transposed conv output size " + std::to_string(out) + " != " + std::to_string(expect);
        return false;
    }
    if (s > kMaxPhase) { *err = "transposed stride > 4"; return false; }
    if (s > 1 && d != 1) { *err = "transposed conv with stride>1 needs dilation 1"; return false; }
    ax->nphase = s; ax->in_step = 1; ax->tap_step = (s == 1) ? d : 1;
    for (int a = 0; a < s; ++a) {
        // o + p - kidx*d = i*s  ->  kidx = r (mod s) with r = (a + p) mod s   (d = 1 when s > 1)
        const int r = (a + p) % s;
        const int nt = (r < k) ? (k - r + s - 1) / s : 0;
        ax->ntap[a] = nt;
        // taps visited in increasing input index: t = 0 is the LARGEST weight index
        ax->kidx0[a] = r + s * (nt - 1);
        ax->kidx_step[a] = -s;
        ax->base[a] = (a + p - r) / s - (nt - 1) * ax->tap_step;
        ax->nout[a] = (a < out) ? (out - a + s - 1) / s : 0;
    }
    return true;
}

struct ConvPlan {
    AxisPlan h, w;
    int Ctot = 0;          // channels summed over (C0+C1 for concat, C0 otherwise)
    int ci_chunk = 2;      // channels staged per LDS pass (even)
    int nchunk = 1;
    int nchunk0 = 1;       // chunks of the first source; a chunk never mixes channels of in0 and in1
    int NB = 64;           // output channels per workgroup (32 or 64)
    int nco_blk = 1;
    int WM = 2, WN = 2;    // wave grid inside the workgroup (WM*WN = 4)
    int NG = 1, TPG = 4;   // per wave: NG column-phase groups x TPG rows of 32x32 accumulator tiles
    int TI = 8;            // phase-index rows per workgroup = WM * TPG
    int nrows = 1, ncols = 1, pitch = 1, plane = 1;   // LDS input tile (floats)
    int w_colmin = 0;      // smallest column offset over the column phases
    int wtaps[kMaxPhase] = {0, 0, 0, 0};      // tap instances per channel pair, per row phase
    size_t w_phase_off[kMaxPhase] = {0, 0, 0, 0};   // float offset of each row phase's image
    size_t packed_floats = 0;
    int in_tile_floats = 0, w_tile_floats = 0;
    size_t lds_bytes = 0, lds_bytes_dma = 0;
    int nit[kMaxPhase] = {0, 0, 0, 0};        // row tiles per row phase
    int njt = 1;                               // column tiles
    // direct (vector-ALU) family for the thin layers; see direct_conv.hip
    bool direct = false;
    int direct_rpt = 8, direct_kt = 1, direct_kpad = 1, direct_pad = 0;
};

constexpr int kSmallGridWgs = 256;      // a grid of at most this many workgroups takes one row tile per wave
constexpr int kTallTileWgs = 600;       // five or six row tiles per wave from this many workgroups (more than the
                                        // chip's 512 resident ones; measured: 600 serves batch 8 better than 2048)

struct ConvPlan;
// kernels with one or two row tiles per wave exist for the LDS-DMA kernels with unrolled tap rows, 64-channel blocks
inline bool conv_one_row_tile_built(const UnetmiConvDesc& d, const ConvPlan& pl);

// Thin layers go to the direct kernel: kernel width 1, unit stride and dilation,
// height 1/3/7, and either <= 2 output channels or (<= 32 outputs from <= 16 inputs).
inline bool plan_direct(const UnetmiConvDesc& d, ConvPlan* pl) {
    if (d.kw != 1 || d.sw != 1 || d.sh != 1 || d.dh != 1 || d.dw != 1 || d.pw != 0 || d.opw || d.oph) return false;
    if (d.kh != 1 && d.kh != 3 && d.kh != 7) return false;
    const int ctot = (d.in_mode == UNETMI_IN_CONCAT) ? d.C0 + d.C1 : d.C0;
    if (d.K <= 2) { pl->direct_rpt = (d.K == 1 && d.kh > 1) ? 16 : 8; pl->direct_kt = d.K; pl->direct_kpad = d.K; }
    // ... or up to 128 outputs from one or two inputs: a pure write stream (the input gradient of a 96 -> 1 layer)
    else if ((d.K <= 32 && ctot * d.kh <= 16 * 7) || (d.K <= 128 && ctot <= 2)) {
        // a thread owns 4 rows x 16 outputs (each output tile re-reads the input, which is the small side here);
        // with at least as many inputs as outputs it owns 2 rows x all 32 outputs, so the input is read once
        // (measured at [32, C, 257, 509]: 64 -> 32 1x1 0.73 -> 0.59 ms, 2 -> 32 7x1 0.204 -> 0.180 ms)
        if (d.K > 16 && ctot >= 32) { pl->direct_rpt = 2; pl->direct_kt = 32; pl->direct_kpad = 32; }
        else { pl->direct_rpt = 4; pl->direct_kt = 16; pl->direct_kpad = (d.K + 15) & ~15; }
    }
    else return false;
    // conv form: ih = oh - pad + kh'; a transposed layer is the flipped kernel with pad' = kh-1-p
    pl->direct_pad = d.transposed ? (d.kh - 1 - d.ph) : d.ph;
    if (pl->direct_pad < 0) return false;
    pl->direct = true;
    return true;
}

inline bool conv_one_row_tile_built(const UnetmiConvDesc& d, const ConvPlan& pl) {
    if (d.reserved[0] == UNETMI_DTYPE_BF16 || d.in_mode == UNETMI_IN_ADD) return false;    // register-staged kernels
    if (pl.w.tap_step != 1 || pl.WN != 2) return false;
    const int t0 = pl.w.ntap[0], t1 = pl.w.ntap[1];
    if (pl.NG == 1) return t0 == 1 || t0 == 3 || t0 == 5 || t0 == 7;
    return t1 == 2 && (t0 == 1 || t0 == 3);
}

inline bool make_plan(const UnetmiConvDesc& d, ConvPlan* pl, std::string* err) {
    if (d.N < 0 || d.C0 < 1 || d.C1 < 0 || d.K < 1) { *err = "bad channel/batch counts"; return false; }
    if (d.in_mode == UNETMI_IN_SINGLE && d.C1 != 0) { *err = "C1 must be 0 for a single input"; return false; }
    if (d.in_mode == UNETMI_IN_ADD && d.C1 != d.C0) { *err = "add mode needs C1 == C0"; return false; }
    if (d.in_mode == UNETMI_IN_CONCAT && d.C1 < 1) { *err = "concat mode needs C1 > 0"; return false; }
    if (d.in_mode < 0 || d.in_mode > 2) { *err = "unknown in_mode"; return false; }
    if (d.act < 0 || d.act > UNETMI_ACT_GELU) { *err = "unknown activation"; return false; }
    if (!d.transposed && (d.oph || d.opw)) { *err = "output_padding on a plain conv"; return false; }
    if (!plan_axis(d.transposed, d.H, d.OH, d.kh, d.sh, d.ph, d.dh, d.oph, &pl->h, err)) return false;
    if (!plan_axis(d.transposed, d.W, d.OW, d.kw, d.sw, d.pw, d.dw, d.opw, &pl->w, err)) return false;
    if (pl->w.nphase != 1 && pl->w.nphase != 2) { *err = "column stride of a transposed conv must be 1 or 2"; return false; }

    pl->Ctot = (d.in_mode == UNETMI_IN_CONCAT) ? d.C0 + d.C1 : d.C0;
    if (plan_direct(d, pl)) {
        pl->packed_floats = std::max<size_t>((size_t)pl->Ctot * d.kh * pl->direct_kpad, 4);
        pl->lds_bytes = 0;
        return true;
    }
    // 64 output channels per workgroup, or 32 when that pads the channel count less (K = 90: 3 x 32 = 96 against
    // 2 x 64 = 128 -- a quarter of the matrix work of a 64-wide second block would be on padding)
    pl->NB = (d.K <= 32 || (d.K + 31) / 32 * 32 < (d.K + 63) / 64 * 64) ? 32 : 64;
    pl->nco_blk = (d.K + pl->NB - 1) / pl->NB;
    pl->WN = pl->NB / 32; pl->WM = kWaves / pl->WN;
    pl->NG = pl->w.nphase; pl->TPG = 4;
    pl->TI = pl->WM * pl->TPG;

    int max_nth = 0;
    for (int a = 0; a < pl->h.nphase; ++a) max_nth = std::max(max_nth, pl->h.ntap[a]);
    pl->nrows = (pl->TI - 1) * pl->h.in_step + std::max(max_nth - 1, 0) * pl->h.tap_step + 1;
    int cmin = 1 << 30, cmax = -(1 << 30), sum_ntw = 0;
    for (int g = 0; g < pl->w.nphase; ++g) {
        sum_ntw += pl->w.ntap[g];
        if (pl->w.ntap[g] == 0) continue;
        cmin = std::min(cmin, pl->w.base[g]);
        cmax = std::max(cmax, pl->w.base[g] + (pl->w.ntap[g] - 1) * pl->w.tap_step);
    }
    if (cmin > cmax) { cmin = 0; cmax = 0; }
    pl->w_colmin = cmin;
    pl->ncols = (kTileJ - 1) * pl->w.in_step + (cmax - cmin) + 1;
    pl->pitch = pl->ncols;          // flat [ci][row][col] image: LDS offset == element index
    pl->plane = pl->nrows * pl->pitch;
    if (max_nth * std::max(pl->w.ntap[0], pl->w.ntap[1]) > kMaxTaps) { *err = "kernel window too large"; return false; }
    if ((long long)d.H * d.W * 4 * std::max(std::max(d.C0, d.C1), 34) >= (1ll << 30)) {
        *err = "image too large for 32-bit buffer offsets"; return false;
    }
    if ((long long)d.K * d.OH * d.OW * 4 >= (1ll << 31)) {      // the epilogue addresses one sample's output with 32-bit offsets
        *err = "output of one sample too large for 32-bit buffer offsets"; return false;
    }

    int max_wtaps = 0;
    for (int a = 0; a < pl->h.nphase; ++a) {
        pl->wtaps[a] = pl->h.ntap[a] * sum_ntw;
        max_wtaps = std::max(max_wtaps, pl->wtaps[a]);
    }
    // channels per LDS pass: the most that fits next to their weights
    const int cpad = (pl->Ctot + 1) & ~1;
    auto in_floats = [&](int prs) { return (2 * prs * pl->plane + 255) & ~255; };
    // + one tap of slack: the MFMA loop reads one weight step past the last tap
    auto w_floats = [&](int prs) { return ((prs * max_wtaps * 2 * pl->NB / 4 + 255) & ~255) * 4 + 2 * pl->NB; };
    int max_pairs = std::min(16, cpad / 2);
    const int budget = (pl->NG == 1 && (in_floats(1) + w_floats(1)) * 4 <= kLdsBudget3) ? kLdsBudget3 : kLdsBudget;
    while (max_pairs > 1 && (in_floats(max_pairs) + w_floats(max_pairs)) * 4 > budget) --max_pairs;
    // what the register-staged kernel can hold per thread while the previous chunk computes
    max_pairs = std::min(max_pairs, std::max(1, kMaxInSlots * 256 / (2 * pl->plane)));
    if (max_wtaps > 0) max_pairs = std::min(max_pairs, std::max(1, kMaxW4Slots * 1024 / (max_wtaps * 2 * pl->NB)));
    if ((in_floats(max_pairs) + w_floats(max_pairs)) * 4 > kLdsBudget || 2 * max_pairs * pl->plane > kMaxInSlots * 256 ||
        max_pairs * max_wtaps * 2 * pl->NB > kMaxW4Slots * 1024) {
        *err = "one channel pair does not fit in LDS for this geometry"; return false;
    }
    // among chunk sizes in (max/2, max], take the one that pads the channel
    // count the least (padding is MFMA work on zeros); ties go to the larger.
    // A concatenated input is chunked per source, so that every chunk streams from one tensor.
    const bool cat = d.in_mode == UNETMI_IN_CONCAT;
    const int prs0 = cat ? (d.C0 + 1) / 2 : cpad / 2, prs1 = cat ? (d.C1 + 1) / 2 : 0;
    int pairs = max_pairs, best_total = 1 << 30;
    for (int p = max_pairs; p >= std::max(1, max_pairs / 2 + 1); --p) {
        const int total = ((prs0 + p - 1) / p + (prs1 + p - 1) / p) * p;
        if (total < best_total) { best_total = total; pairs = p; }
    }
    pl->ci_chunk = 2 * pairs;
    pl->nchunk0 = (prs0 + pairs - 1) / pairs;
    pl->nchunk = pl->nchunk0 + (prs1 + pairs - 1) / pairs;

    size_t off = 0;
    for (int a = 0; a < pl->h.nphase; ++a) {
        pl->w_phase_off[a] = off;
        off += (size_t)pl->nco_blk * pl->nchunk * pairs * pl->wtaps[a] * 2 * pl->NB;
    }
    pl->packed_floats = std::max<size_t>(off, 4);
    // both LDS images are padded to whole 256-thread slots so staging needs no per-lane guard
    pl->in_tile_floats = in_floats(pairs);
    pl->w_tile_floats = w_floats(pairs);
    pl->lds_bytes = (size_t)(pl->in_tile_floats + pl->w_tile_floats + 5 * pl->NB) * 4;          // register-staged kernel
    pl->lds_bytes_dma = (size_t)(2 * (pl->in_tile_floats + pl->w_tile_floats) + 5 * pl->NB) * 4; // double-buffered kernel

    int max_nj = 0;
    for (int g = 0; g < pl->w.nphase; ++g) max_nj = std::max(max_nj, pl->w.nout[g]);
    pl->njt = (max_nj + kTileJ - 1) / kTileJ;
    for (int a = 0; a < pl->h.nphase; ++a) pl->nit[a] = (pl->h.nout[a] + pl->TI - 1) / pl->TI;

    // Row tiles per wave.  Small grids (the deep 3x32 ... 9x64 maps): with at most one workgroup per CU a launch
    // lasts as long as ONE wave's serial walk over all channels and taps with its four row tiles, whatever the FLOP
    // count.  One row tile per wave (two output rows per workgroup) gives four times the workgroups and a quarter
    // of that chain.
    // The channel chunking -- and with it the packed weight image -- stays as computed above for four row tiles,
    // so one packed image serves every batch size.
    long long wgs = 0;
    for (int a = 0; a < pl->h.nphase; ++a) wgs += (long long)pl->nit[a] * pl->njt * d.N * pl->nco_blk;
    // reserved[1] of the descriptor: 0 = decide from the grid, 4 / 2 / 1 = that many row tiles where built.
    // Mid-sized grids: the launch takes ceil(workgroups / resident slots) rounds, so 1.4 rounds of work cost 2;
    // two row tiles per wave halve the granule (at ~8 % more LDS reads and weight traffic per FLOP) and are taken
    // when that saves more than it costs.
    const bool built = conv_one_row_tile_built(d, *pl);
    int tpg = 4;
    if (built && d.reserved[1] == 1) tpg = 1;
    else if (built && d.reserved[1] == 2) tpg = 2;
    else if (built && d.reserved[1] == 0) {
        if (wgs <= kSmallGridWgs) tpg = 1;
        else {
            const long long slots = (pl->NG == 1 ? 3 : 2) * 256;          // resident workgroups of the chip
            long long wgs2 = 0;
            for (int a = 0; a < pl->h.nphase; ++a)
                wgs2 += (long long)((pl->h.nout[a] + 2 * pl->WM - 1) / (2 * pl->WM)) * pl->njt * d.N * pl->nco_blk;
            const long long r4 = (wgs + slots - 1) / slots, r2 = (wgs2 + slots - 1) / slots;
            if (r2 * 2 * 108 < r4 * 4 * 95) tpg = 2;
        }
    }
    // Five or six row tiles per wave (ten or twelve output rows per workgroup, 80 / 96 accumulators, two workgroups
    // per CU instead of three): every workgroup fetches the whole weight tensor from L2 through its CU's vector L1,
    // whose request queue is what the large layers wait for, and a taller tile fetches fewer of them -- and of the
    // halo rows -- per output row.  Taken on large grids for the height that pads the rows least, when that is no
    // more than eight-row tiles pad them (257 rows: 260 in tens, 264 in eights or twelves).  The channel chunking
    // stays as computed above (same packed image).
    // With two column phases (twice the accumulators) five row tiles are the most a wave's 256 registers hold.
    if (tpg == 4 && built && (d.reserved[1] == 5 || (d.reserved[1] == 6 && pl->NG == 1) || d.reserved[1] == 0)) {
        auto padded = [&](int ti) {
            long long rows = 0;
            for (int a = 0; a < pl->h.nphase; ++a) rows += (long long)((pl->h.nout[a] + ti - 1) / ti) * ti;
            return rows;
        };
        int tall = d.reserved[1];
        if (tall == 0) tall = (pl->NG == 2 || padded(10) < padded(12)) ? 5 : 6;
        long long wgs_tall = 0;
        for (int a = 0; a < pl->h.nphase; ++a)
            wgs_tall += (long long)((pl->h.nout[a] + 2 * tall - 1) / (2 * tall)) * pl->njt * d.N * pl->nco_blk;
        const int nrows_was = pl->nrows;
        pl->nrows = (2 * tall - 1) * pl->h.in_step + std::max(max_nth - 1, 0) * pl->h.tap_step + 1;
        pl->plane = pl->nrows * pl->pitch;
        const bool fits = (size_t)(2 * (in_floats(pairs) + w_floats(pairs)) + 5 * pl->NB) * 4 <= 80 * 1024;
        if (fits && (d.reserved[1] != 0 || (wgs_tall >= kTallTileWgs && padded(2 * tall) <= padded(8)))) {
            pl->TPG = tall; pl->TI = 2 * tall;
            pl->in_tile_floats = in_floats(pairs);
            pl->lds_bytes = 0;                                   // no register-staged kernel of this shape
            pl->lds_bytes_dma = (size_t)(2 * (pl->in_tile_floats + pl->w_tile_floats) + 5 * pl->NB) * 4;
            for (int a = 0; a < pl->h.nphase; ++a) pl->nit[a] = (pl->h.nout[a] + pl->TI - 1) / pl->TI;
        } else {
            pl->nrows = nrows_was; pl->plane = pl->nrows * pl->pitch;
        }
    }
    if (tpg != 4) {
        pl->TPG = tpg;
        pl->TI = pl->WM * tpg;
        pl->nrows = (pl->TI - 1) * pl->h.in_step + std::max(max_nth - 1, 0) * pl->h.tap_step + 1;
        pl->plane = pl->nrows * pl->pitch;
        // A chunk of a one-row-tile workgroup is a few hundred MFMAs, shorter than the round trip of the next
        // chunk's DMA, so the launch would last (chunks x round trip).  Merge k consecutive chunks while both LDS
        // images stay within 80 KB: with one column phase the weight image is contiguous in channel pairs, so it
        // does not change as long as k divides the chunk count of each source (with two column phases a chunk
        // holds phase 0's pairs and then phase 1's, so those layers keep their chunks).
        for (int k = (tpg == 1 && pl->w.nphase == 1) ? 8 : 0; k >= 2; k /= 2) {
            const int n1 = pl->nchunk - pl->nchunk0;
            if (pl->nchunk0 % k || n1 % k) continue;
            if ((size_t)(2 * (in_floats(pairs * k) + w_floats(pairs * k)) + 5 * pl->NB) * 4 > 80 * 1024) continue;
            pairs *= k;
            pl->ci_chunk = 2 * pairs; pl->nchunk0 /= k; pl->nchunk /= k;
            pl->w_tile_floats = w_floats(pairs);
            break;
        }
        pl->in_tile_floats = in_floats(pairs);
        pl->lds_bytes = (size_t)(pl->in_tile_floats + pl->w_tile_floats + 5 * pl->NB) * 4;
        pl->lds_bytes_dma = (size_t)(2 * (pl->in_tile_floats + pl->w_tile_floats) + 5 * pl->NB) * 4;
        for (int a = 0; a < pl->h.nphase; ++a) pl->nit[a] = (pl->h.nout[a] + pl->TI - 1) / pl->TI;
    }
    return true;
}

// Two loads per input element are needed for the skip add only.
inline bool needs_two_loads(const UnetmiConvDesc& d, const ConvPlan&) {
    return d.in_mode == UNETMI_IN_ADD;
}

// Channel of the (concatenated) input that position i of chunk `chunk` holds, or -1 for padding.
inline int chunk_channel(const UnetmiConvDesc& d, const ConvPlan& pl, int chunk, int i) {
    if (d.in_mode != UNETMI_IN_CONCAT) {
        const int c = chunk * pl.ci_chunk + i;
        return c < pl.Ctot ? c : -1;
    }
    if (chunk < pl.nchunk0) {
        const int c = chunk * pl.ci_chunk + i;
        return c < d.C0 ? c : -1;
    }
    const int c = (chunk - pl.nchunk0) * pl.ci_chunk + i;
    return c < d.C1 ? d.C0 + c : -1;
}

}  // namespace unetmi
