// Developer probe: times one fused-conv layer of libunetmi's kernel with parts of
// it switched off (ablation), to see where a launch spends its time.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DUNETMI_ABLATE tools/conv_probe.hip -o gpurun_out/conv_probe
//   conv_probe N C0 C1 H W K kh kw sh sw ph pw transposed [oph opw]
// Ablation bits: 1 = no re-staging after the first chunk, 2 = no MFMA, 4 = no output stores,
//                8 = no global prefetch loads after the first chunk.
#include "../unet_amd/csrc/conv_mfma.hip"
namespace unetmi { extern int g_xcd_remap; }

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" int unetmi_conv2d_pack_weights(const UnetmiConvDesc*, const float*, float*);
#include "../unet_amd/csrc/unetmi_api.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    if (argc < 14) { printf("usage\n"); return 1; }
    UnetmiConvDesc d{};
    int* f[] = {&d.N,&d.C0,&d.C1,&d.H,&d.W,&d.K,&d.kh,&d.kw,&d.sh,&d.sw,&d.ph,&d.pw,&d.transposed};
    for (int i = 0; i < 13; ++i) *f[i] = atoi(argv[i + 1]);
    if (argc >= 16) { d.oph = atoi(argv[14]); d.opw = atoi(argv[15]); }
    d.dh = d.dw = 1; d.in_mode = d.C1 ? UNETMI_IN_CONCAT : UNETMI_IN_SINGLE; d.act = UNETMI_ACT_LEAKY;
    if (d.transposed) { d.OH = (d.H-1)*d.sh - 2*d.ph + d.kh + d.oph; d.OW = (d.W-1)*d.sw - 2*d.pw + d.kw + d.opw; }
    else { d.OH = (d.H + 2*d.ph - d.kh)/d.sh + 1; d.OW = (d.W + 2*d.pw - d.kw)/d.sw + 1; }
    unetmi::ConvPlan pl; std::string err;
    if (!unetmi::make_plan(d, &pl, &err)) { printf("plan: %s\n", err.c_str()); return 1; }
    const int C = d.C0 + d.C1;
    printf("plan: ci_chunk %d nchunk %d NB %d TI %d nrows %d ncols %d lds %zu B, in slots %d w4 slots %d\n",
           pl.ci_chunk, pl.nchunk, pl.NB, pl.TI, pl.nrows, pl.ncols, pl.lds_bytes,
           (pl.ci_chunk * pl.plane + 255) / 256, (pl.ci_chunk/2 * pl.wtaps[0] * 2 * pl.NB / 4 + 255) / 256);
    size_t n0 = (size_t)d.N*d.C0*d.H*d.W, n1 = (size_t)d.N*d.C1*d.H*d.W, no = (size_t)d.N*d.K*d.OH*d.OW;
    std::vector<float> hw((size_t)C*d.K*d.kh*d.kw), hp(pl.packed_floats), hin(n0 + n1 + 16);
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-4f;
    for (auto& v : hin) v = (rand() % 2001 - 1000) * 1e-3f;
    unetmi_conv2d_pack_weights(&d, hw.data(), hp.data());
    float *in0, *in1, *wp, *par, *out;
    CK(hipMalloc(&in0, (n0 + 16)*4)); CK(hipMalloc(&in1, (n1 + 16)*4)); CK(hipMalloc(&wp, hp.size()*4));
    CK(hipMalloc(&par, d.K*4*4)); CK(hipMalloc(&out, no*4));
    CK(hipMemcpy(in0, hin.data(), n0*4, hipMemcpyHostToDevice));
    if (n1) CK(hipMemcpy(in1, hin.data() + n0, n1*4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wp, hp.data(), hp.size()*4, hipMemcpyHostToDevice));
    std::vector<float> ones(d.K*4, 1.f); for (int i = 0; i < d.K; ++i) ones[3*d.K + i] = 0.01f;
    CK(hipMemcpy(par, ones.data(), d.K*4*4, hipMemcpyHostToDevice));
    const double macs = d.transposed ? (double)d.N*d.H*d.W*C*d.K*d.kh*d.kw : (double)d.N*d.OH*d.OW*d.K*C*d.kh*d.kw;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int flagsets[] = {0, 0, 8, 16, 16|32, 4, 4|16|32, 2};
    printf("bits: 8 DMA out of range after chunk 0, 16 no DMA issue after chunk 0, 32 no chunk barrier, 4 no stores, 2 no MFMA\n");
    int idx = 0;
    for (int fl : flagsets) {
        unetmi::g_ablate = fl;
        unetmi::g_xcd_remap = 0; (void)idx;
        printf("xcd_remap %d ", unetmi::g_xcd_remap);
        for (int it = 0; it < 2; ++it)
            CK(unetmi::launch_conv_mfma(d, pl, in0, n1 ? in1 : nullptr, wp, par, par + d.K, par + 2*d.K, par + 3*d.K, nullptr, nullptr, out, 0));
        CK(hipEventRecord(e0, 0));
        const int iters = 5;
        for (int it = 0; it < iters; ++it)
            CK(unetmi::launch_conv_mfma(d, pl, in0, n1 ? in1 : nullptr, wp, par, par + d.K, par + 2*d.K, par + 3*d.K, nullptr, nullptr, out, 0));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
        printf("ablate %2d: %8.3f ms  %7.1f TFLOP/s-equivalent\n", fl, ms, 2e-9 * macs / ms);
    }
    return 0;
}
