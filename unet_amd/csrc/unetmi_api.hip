// C ABI of libunetmi.so (declarations and contracts: include/unetmi.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "conv_plan.hpp"

namespace unetmi {
hipError_t launch_conv_mfma(const UnetmiConvDesc& d, const ConvPlan& pl,
                            const float* in0, const float* in1, const float* wpacked,
                            const float* bias, const float* scale, const float* shift,
                            const float* act_param, const float* gate, const float* residual,
                            float* out, hipStream_t stream, double* stats, int stats_S);
int conv_stats_slots(const UnetmiConvDesc& d, const ConvPlan& pl);
int launch_bn_stats_finish(const double* ws, int C, int S, long long M, float eps, float* mean, float* invstd, hipStream_t s);

hipError_t launch_direct_conv(const UnetmiConvDesc& d, const ConvPlan& pl,
                              const float* in0, const float* in1, const float* wpacked,
                              const float* bias, const float* scale, const float* shift,
                              const float* act_param, const float* gate, const float* residual,
                              float* out, hipStream_t stream);

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int api_fail(int code, const std::string& msg) { return fail(code, msg); }   // for the other translation units
}  // namespace unetmi

using namespace unetmi;

extern "C" {

int unetmi_version(void) { return UNETMI_VERSION; }

void unetmi_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return;
    std::strncpy(buf, g_last_error.c_str(), n - 1);
    buf[n - 1] = '\0';
}

size_t unetmi_conv2d_packed_weight_bytes(const UnetmiConvDesc* d) {
    if (!d) { fail(UNETMI_E_BAD_ARG, "null descriptor"); return 0; }
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) { fail(UNETMI_E_BAD_DESC, err); return 0; }
    return pl.packed_floats * sizeof(float);
}

int unetmi_conv2d_kernel_name(const UnetmiConvDesc* d, char* buf, size_t n) {
    if (!d || !buf || n == 0) return fail(UNETMI_E_BAD_ARG, "null argument to conv2d_kernel_name");
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) return fail(UNETMI_E_BAD_DESC, err);
    char name[96];
    if (pl.direct) {
        std::snprintf(name, sizeof(name), "direct_conv_kernel<%d, %d, %d>", pl.direct_rpt, pl.direct_kt, d->kh);
    } else if (d->reserved[0] == UNETMI_DTYPE_BF16 || needs_two_loads(*d, pl)) {      // the register-staged kernel
        std::snprintf(name, sizeof(name), "conv_mfma_kernel<%d, 4, %d, %d, %s, 0, 0>", pl.NG, pl.WM, pl.WN,
                      needs_two_loads(*d, pl) ? "true" : "false");
    } else {                                                  // same order of preference as launch_conv_mfma
        int t0 = 0, t1 = 0;
        if (pl.w.tap_step == 1) {
            const int a = pl.w.ntap[0], b = pl.w.ntap[1];
            if (pl.NG == 1 && (a == 1 || a == 3 || a == 5 || a == 7)) t0 = a;
            else if (pl.NG == 2 && b == 2 && (a == 1 || a == 3)) { t0 = a; t1 = b; }
        }
        std::snprintf(name, sizeof(name), "conv_dma_kernel<%d, %d, %d, %d, %d, %d>", pl.NG, t0 ? pl.TPG : 4, pl.WM, pl.WN,
                      t0, t1);
    }
    std::snprintf(buf, n, "%s", name);
    return UNETMI_OK;
}

int unetmi_conv2d_pack_weights(const UnetmiConvDesc* d, const float* w, float* packed) {
    if (!d || !w || !packed) return fail(UNETMI_E_BAD_ARG, "null argument to pack_weights");
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) return fail(UNETMI_E_BAD_DESC, err);
    std::memset(packed, 0, pl.packed_floats * sizeof(float));
    const int cpc = pl.ci_chunk / 2;
    const int C = pl.Ctot, K = d->K, KH = d->kh, KW = d->kw;
    if (pl.direct) {            // image [ci][kh'][kpad]; transposed layers flipped along kh
        for (int ci = 0; ci < C; ++ci)
            for (int kh = 0; kh < KH; ++kh)
                for (int co = 0; co < K; ++co) {
                    const size_t src = d->transposed ? (((size_t)ci * K + co) * KH + (KH - 1 - kh))
                                                     : (((size_t)co * C + ci) * KH + kh);
                    packed[((size_t)ci * KH + kh) * pl.direct_kpad + co] = w[src];
                }
        return UNETMI_OK;
    }
    for (int a = 0; a < pl.h.nphase; ++a) {
        float* dst = packed + pl.w_phase_off[a];
        for (int cb = 0; cb < pl.nco_blk; ++cb)
            for (int chunk = 0; chunk < pl.nchunk; ++chunk)
                for (int g = 0; g < pl.w.nphase; ++g)       // image order == kernel read order
                    for (int cp = 0; cp < cpc; ++cp)
                        for (int th = 0; th < pl.h.ntap[a]; ++th)
                            for (int tw = 0; tw < pl.w.ntap[g]; ++tw) {
                                const int ky = pl.h.kidx0[a] + th * pl.h.kidx_step[a];
                                const int kx = pl.w.kidx0[g] + tw * pl.w.kidx_step[g];
                                for (int hf = 0; hf < 2; ++hf) {
                                    const int ci = chunk_channel(*d, pl, chunk, cp * 2 + hf);
                                    for (int nb = 0; nb < pl.NB; ++nb, ++dst) {
                                        const int co = cb * pl.NB + nb;
                                        if (ci < 0 || co >= K) continue;
                                        const size_t src = d->transposed
                                            ? (((size_t)ci * K + co) * KH + ky) * KW + kx
                                            : (((size_t)co * C + ci) * KH + ky) * KW + kx;
                                        *dst = w[src];
                                    }
                                }
                            }
    }
    return UNETMI_OK;
}

int unetmi_conv2d_fwd(const UnetmiConvDesc* d, const float* in0, const float* in1,
                      const float* wpacked, const float* bias, const float* scale,
                      const float* shift, const float* act_param, const float* gate,
                      const float* residual, float* out, void* stream) {
    if (!d) return fail(UNETMI_E_BAD_ARG, "null descriptor");
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) return fail(UNETMI_E_BAD_DESC, err);
    if (d->N == 0) return UNETMI_OK;   // empty batch: nothing to read or write
    if (!in0 || !wpacked || !bias || !scale || !shift || !out)
        return fail(UNETMI_E_BAD_ARG, "null tensor pointer");
    if (d->in_mode != UNETMI_IN_SINGLE && !in1) return fail(UNETMI_E_BAD_ARG, "in_mode needs in1");
    if (d->act == UNETMI_ACT_LEAKY && !act_param) return fail(UNETMI_E_BAD_ARG, "leaky/prelu needs act_param");
    if ((d->has_gate != 0) != (gate != nullptr)) return fail(UNETMI_E_BAD_ARG, "gate pointer vs has_gate");
    if ((d->has_residual != 0) != (residual != nullptr))
        return fail(UNETMI_E_BAD_ARG, "residual pointer vs has_residual");
    if (pl.lds_bytes_dma > 80 * 1024) return fail(UNETMI_E_UNSUPPORTED, "LDS images exceed 80 KB");
    if (d->reserved[0] != UNETMI_DTYPE_F32 && d->reserved[0] != UNETMI_DTYPE_BF16)
        return fail(UNETMI_E_BAD_DESC, "unknown activation storage type in reserved[0]");
    if (d->reserved[1] < 0 || d->reserved[1] > 6 || d->reserved[1] == 3)
        return fail(UNETMI_E_BAD_DESC, "reserved[1] (row tiles per wave) must be 0, 1, 2, 4, 5 or 6");
    if (!pl.direct && pl.NG == 2 && d->reserved[0] == UNETMI_DTYPE_BF16)
        return fail(UNETMI_E_UNSUPPORTED,
                    "bfloat16 storage on a time-stride-2 transposed conv is not built: widen the operands");
    if (!pl.direct && pl.NG == 2 && needs_two_loads(*d, pl))
        return fail(UNETMI_E_UNSUPPORTED,
                    "summed two-source input on a time-stride-2 transposed conv: add the inputs first");
    const hipError_t e = pl.direct ? launch_direct_conv(*d, pl, in0, in1, wpacked, bias, scale, shift, act_param, gate, residual, out,
                                                        static_cast<hipStream_t>(stream))
                                   : launch_conv_mfma(*d, pl, in0, in1, wpacked, bias, scale, shift, act_param, gate, residual, out,
                                                      static_cast<hipStream_t>(stream), nullptr, 0);
    if (e != hipSuccess) return fail(UNETMI_E_HIP, std::string("conv launch: ") + hipGetErrorString(e));
    return UNETMI_OK;
}

size_t unetmi_conv2d_fwd_stats_workspace_bytes(const UnetmiConvDesc* d) {
    if (!d) { fail(UNETMI_E_BAD_ARG, "null descriptor"); return 0; }
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) { fail(UNETMI_E_BAD_DESC, err); return 0; }
    if (pl.direct || d->reserved[0] != UNETMI_DTYPE_F32 || d->act != UNETMI_ACT_NONE || d->has_gate || d->has_residual) {
        fail(UNETMI_E_UNSUPPORTED, "conv2d_fwd_stats: an fp32 layer of the matrix kernels without activation, gate or added tensor");
        return 0;
    }
    return (size_t)d->K * (size_t)conv_stats_slots(*d, pl) * 2 * sizeof(double);
}

int unetmi_conv2d_fwd_stats(const UnetmiConvDesc* d, const float* in0, const float* in1, const float* wpacked,
                            const float* bias, const float* scale, const float* shift, float* out, float eps,
                            float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream) {
    if (!d) return fail(UNETMI_E_BAD_ARG, "null descriptor");
    const size_t need = unetmi_conv2d_fwd_stats_workspace_bytes(d);
    if (need == 0) return d->N == 0 ? fail(UNETMI_E_BAD_DESC, "conv2d_fwd_stats: empty batch") : UNETMI_E_UNSUPPORTED;
    ConvPlan pl;
    std::string err;
    if (!make_plan(*d, &pl, &err)) return fail(UNETMI_E_BAD_DESC, err);
    if (!in0 || !wpacked || !bias || !scale || !shift || !out || !mean || !invstd || !ws)
        return fail(UNETMI_E_BAD_ARG, "conv2d_fwd_stats: null pointer");
    if (d->in_mode != UNETMI_IN_SINGLE && !in1) return fail(UNETMI_E_BAD_ARG, "in_mode needs in1");
    if (ws_bytes < need) return fail(UNETMI_E_BAD_ARG, "conv2d_fwd_stats: workspace too small");
    if (pl.lds_bytes_dma > 80 * 1024) return fail(UNETMI_E_UNSUPPORTED, "LDS images exceed 80 KB");
    if (d->reserved[1] < 0 || d->reserved[1] > 6 || d->reserved[1] == 3)
        return fail(UNETMI_E_BAD_DESC, "reserved[1] (row tiles per wave) must be 0, 1, 2, 4, 5 or 6");
    if (pl.NG == 2 && needs_two_loads(*d, pl))
        return fail(UNETMI_E_UNSUPPORTED, "summed two-source input on a time-stride-2 transposed conv: add the inputs first");
    const int S = conv_stats_slots(*d, pl);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const hipError_t e = launch_conv_mfma(*d, pl, in0, in1, wpacked, bias, scale, shift, nullptr, nullptr, nullptr, out, s,
                                          static_cast<double*>(ws), S);
    if (e != hipSuccess) return fail(UNETMI_E_HIP, std::string("conv launch: ") + hipGetErrorString(e));
    return launch_bn_stats_finish(static_cast<const double*>(ws), d->K, S, (long long)d->N * d->OH * d->OW, eps, mean, invstd, s);
}

}  // extern "C"
