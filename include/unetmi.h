/*
 * unetmi.h -- C ABI of libunetmi.so, the MI355X (gfx950) op library behind the
 * U-Net family's forward pass.
 *
 * Every entry point takes plain device/host pointers and sizes (no torch
 * types), enqueues on the caller's HIP stream, never synchronises the device
 * and never allocates device memory: activations, packed weights and
 * workspaces are owned by the caller.  Return value 0 = success, negative =
 * UNETMI_E_* ; unetmi_last_error() gives the message of the calling thread's
 * last failure.  All tensors are NCHW contiguous, [B, C, F, T] with the time axis innermost,
 * fp32 unless a call says bfloat16 storage (UNETMI_DTYPE_BF16 below) -- the layout of the reference's modules
 * (UNet.py:75, cRFUNet.py:280, ResUNet.py:162).
 *
 * Each function names the reference call sites it replaces.  The reference has
 * no native interface of its own (it is pure torch.nn); the Python host side
 * in unet_amd/ binds these symbols with ctypes and registers them as
 * torch.library ops.
 */
#ifndef UNETMI_H
#define UNETMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNETMI_VERSION 100

enum {
    UNETMI_OK = 0,
    UNETMI_E_BAD_DESC = -1,     /* inconsistent or unsupported geometry      */
    UNETMI_E_BAD_ARG = -2,      /* null pointer / size mismatch              */
    UNETMI_E_HIP = -3,          /* a HIP runtime call failed                 */
    UNETMI_E_UNSUPPORTED = -4   /* valid request this build cannot serve     */
};

/* Epilogue activations (the reference's string switch, UNet_m.py:324-337,
 * plus the heads' Sigmoid / Softplus / Tanh: UNet.py:72, cRFUNet.py:269-272,
 * ResUNet.py:144, DSUNet.py:356). */
enum {
    UNETMI_ACT_NONE = 0,
    UNETMI_ACT_LEAKY = 1,    /* x>=0 ? x : slope*x ; slope = act_param[0] (LeakyReLU 0.01) or
                                act_param[c] (PReLU, see act_param_per_channel) */
    UNETMI_ACT_RELU = 2,
    UNETMI_ACT_SILU = 3,
    UNETMI_ACT_SOFTPLUS = 4, /* beta 1, threshold 20 */
    UNETMI_ACT_SIGMOID = 5,
    UNETMI_ACT_TANH = 6,
    UNETMI_ACT_GELU = 7      /* erf form */
};

/* Storage type of the ACTIVATION tensors of a call (inputs, residual, output).  Arithmetic is fp32 in
 * both cases; parameters, folded BatchNorm vectors and packed weights are always fp32.  With
 * UNETMI_DTYPE_BF16 the tensors hold bfloat16 (widened exactly on load, rounded to nearest-even on
 * store) and the pointers below, typed for the fp32 case, address 2-byte elements. */
enum {
    UNETMI_DTYPE_F32 = 0,
    UNETMI_DTYPE_BF16 = 1,
    UNETMI_DTYPE_F16 = 2        /* blocked 16-bit tensors of the unetmi_conv2d_b16_* family only */
};

/* How the second input tensor combines with the first. */
enum {
    UNETMI_IN_SINGLE = 0,
    UNETMI_IN_CONCAT = 1,    /* channels [0,C0) from in0, [C0,C0+C1) from in1: torch.cat([p, skip], 1)
                                fused into the read (UNet.py:106, cRFUNet.py:311, DSUNet.py:622) */
    UNETMI_IN_ADD = 2        /* in0 + in1 elementwise, C1 == C0: torch.add(y, res) fused into the
                                read (ResUNet.py:209) */
};

/* Geometry + epilogue of one fused convolution layer.
 *
 * transposed = 0: nn.Conv2d; weight [K, (C0+C1)/groups, kh, kw].
 * transposed = 1: nn.ConvTranspose2d; weight [C0+C1, K, kh, kw]; stride > 1
 *                 requires dilation 1 on that axis.
 * groups is 1 (dense) for this entry point.
 * Epilogue, per output element v = acc + bias[k]:
 *   act_first = 0:  v = act(v * scale[k] + shift[k])      (conv -> BN -> act:  UNet_m.py:340-345, :379-383)
 *   act_first = 1:  v = act(v) * scale[k] + shift[k]      (conv -> act -> BN:  UNet_m.py:549-552)
 *   then v *= gate[n*K + k] if has_gate                    (DSUNet.py:599-602)
 *   then v += residual[n,k,oh,ow] if has_residual          (UNet_m.py:561)
 * scale/shift are eval-mode BatchNorm folded by the caller:
 *   scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale.
 */
typedef struct UnetmiConvDesc {
    int32_t N;               /* batch                                        */
    int32_t C0, C1;          /* channels of in0 and in1 (C1 = 0 if single)   */
    int32_t H, W;            /* input  F, T                                  */
    int32_t K;               /* output channels                              */
    int32_t OH, OW;          /* output F, T (must equal the torch formula)   */
    int32_t kh, kw;
    int32_t sh, sw;
    int32_t ph, pw;
    int32_t dh, dw;
    int32_t oph, opw;        /* output_padding (transposed only)             */
    int32_t transposed;
    int32_t in_mode;         /* UNETMI_IN_*                                  */
    int32_t act;             /* UNETMI_ACT_*                                 */
    int32_t act_first;
    int32_t act_param_per_channel; /* 1: act_param has K entries, 0: one     */
    int32_t has_gate;
    int32_t has_residual;
    int32_t reserved[3];     /* reserved[0] = UNETMI_DTYPE_* of in0 / in1 / residual / out;
                                unetmi_conv2d_fwd: reserved[1] = row tiles per wave of the MFMA kernels -- 0: chosen from
                                the grid size (one for grids of at most 256 workgroups, two where that saves a round
                                of workgroups, else four; large grids of single-column-phase layers take five or six when
                                ten- or twelve-row tiles pad the rows no more than eight-row tiles), 4: always four,
                                1 / 2 / 5 / 6: that many wherever the kernel is built; the result does not depend on it.  reserved[2] = 0.
                                unetmi_conv2d_b16_fwd: [1], [2] as described there */
} UnetmiConvDesc;

int unetmi_version(void);

/* Copies the calling thread's last error message (NUL-terminated) into buf. */
void unetmi_last_error(char* buf, size_t n);

/* Bytes of the packed-weight image unetmi_conv2d_fwd expects for this layer
 * (0 on a bad descriptor).  The image depends on geometry only, not on N. */
size_t unetmi_conv2d_packed_weight_bytes(const UnetmiConvDesc* d);

/* HOST function, a measurement aid with no counterpart in the reference: the kernel instantiation
 * unetmi_conv2d_fwd launches for this descriptor (it depends on N through the grid size), written as a profiler
 * shows it without namespace, storage-type parameter and argument list, e.g. "conv_dma_kernel<1, 4, 2, 2, 5, 0>".
 * bench.py labels its per-launch timings with it so that they can be matched against a rocprofv3 kernel trace. */
int unetmi_conv2d_kernel_name(const UnetmiConvDesc* d, char* buf, size_t n);

/* HOST function: rearranges a torch-layout weight tensor (host pointer,
 * contiguous fp32, [K,C,kh,kw] or [C,K,kh,kw] for transposed) into the
 * per-phase / per-channel-chunk image the kernel streams through LDS
 * (host pointer, unetmi_conv2d_packed_weight_bytes() bytes).  Called once per
 * layer when parameters change; the caller uploads the image. */
int unetmi_conv2d_pack_weights(const UnetmiConvDesc* d, const float* w, float* packed);

/* Fused convolution layer on the device.  Replaces, in one launch,
 *   Encoder.forward  conv -> dropout(p=0) -> bn -> acti       UNet_m.py:340-345
 *   Decoder.forward  transconv -> bn -> acti                  UNet_m.py:379-383
 *   the torch.cat feeding each decoder                        UNet.py:106
 *   the 1x1 heads (linear + Sigmoid / Softplus)               UNet.py:110-111, cRFUNet.py:315-316
 *   ResUNet's res_path 1x1 blocks and y + res                 ResUNet.py:125-132, :209
 *   DSUNet's per-(batch,channel) gate                         DSUNet.py:599-602
 * in0/in1/out/residual: device fp32 NCHW; wpacked: device image from
 * unetmi_conv2d_pack_weights; bias/scale/shift: device fp32 [K];
 * act_param: device fp32 [1] or [K] (may be null for activations without a
 * parameter); gate: device fp32 [N*K] or null; residual: device fp32 shaped
 * like out, or null.  stream: hipStream_t passed as void*. */
int unetmi_conv2d_fwd(const UnetmiConvDesc* d,
                      const float* in0, const float* in1,
                      const float* wpacked,
                      const float* bias, const float* scale, const float* shift,
                      const float* act_param, const float* gate,
                      const float* residual,
                      float* out, void* stream);

/* Training-time form of unetmi_conv2d_fwd for a layer followed by BatchNorm2d with batch statistics
 * (UNet_m.py:253-256 / :317-320 in train() mode: nn.Conv2d -> nn.BatchNorm2d): the convolution (bias, and the affine
 * given by scale / shift; no activation, gate or added tensor: d->act = NONE, has_gate = has_residual = 0) whose
 * epilogue also leaves per-tile partial sums (sum z, sum z^2) of its output, followed by one small launch that turns
 * them into mean[K] and invstd[K] = 1 / sqrt(biased variance + eps) -- the statistics cost no pass over the tensor.
 * The partials are added in a fixed order (lanes by a butterfly, waves and tiles in index order, in double), so the
 * result does not depend on launch order.  ws: unetmi_conv2d_fwd_stats_workspace_bytes(d) bytes of device scratch
 * (K x tiles x 2 doubles, every one written by the call); that function returns 0 -- and this one
 * UNETMI_E_UNSUPPORTED -- for a layer the matrix kernels do not serve (thin layers on the streaming kernel,
 * bfloat16 storage): run unetmi_conv2d_fwd and unetmi_bn_train_stats there. */
size_t unetmi_conv2d_fwd_stats_workspace_bytes(const UnetmiConvDesc* d);
int unetmi_conv2d_fwd_stats(const UnetmiConvDesc* d, const float* in0, const float* in1, const float* wpacked,
                            const float* bias, const float* scale, const float* shift, float* out, float eps,
                            float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream);

/* Depthwise (groups = C) convolution, stride 1, 'same' padding 2p = d(k-1); kernel 3x3,
 * 1x3 or 3x1.  Replaces the first conv of each ResBlock layer
 *   nn.Conv2d(C, C, (3,3), (1,1), (1,d), (1,d), groups=C)            UNet_m.py:546-548
 * and, with in2, the torch.add(y, res) in front of a decoder's ResBlock       ResUNet.py:209
 * in/in2/out: device fp32 [N,C,H,W] (in2 null, or an addend summed into the input on the fly);
 * weight: device fp32 [C,1,kh,kw] (torch layout); bias: [C] or null. */
int unetmi_dwconv2d_fwd(int N, int C, int H, int W, int kh, int kw, int ph, int pw, int dh, int dw,
                        const void* in, const void* in2, const float* weight, const float* bias, void* out,
                        int dtype, void* stream);

/* Weight and bias gradient of that depthwise convolution for train mode (autograd through UNet_m.py:546-548):
 *   dweight[c][0][r][s] = sum_{n,y,x} dout[n,c,y,x] * (in [+ in2])[n,c, y-ph+r*dh, x-pw+s*dw],  dbias[c] = sum dout
 * in/in2/dout: device fp32 [N,C,H,W]; dweight [C,1,kh,kw]; dbias [C] or null; ws: device scratch of
 * unetmi_dwconv2d_wgrad_workspace_bytes(C, kh, kw) bytes.  Sums are formed in a fixed order.  The input gradient
 * is unetmi_dwconv2d_fwd applied to dout with the taps of each channel reversed. */
size_t unetmi_dwconv2d_wgrad_workspace_bytes(int C, int kh, int kw);
int unetmi_dwconv2d_wgrad(int N, int C, int H, int W, int kh, int kw, int ph, int pw, int dh, int dw, const float* in,
                          const float* in2, const float* dout, float* dweight, float* dbias, void* ws, size_t ws_bytes,
                          void* stream);

/* One layer of UNet_m.ResBlock, fused (C = 64):
 *   out = [x (+ x2) +] BN(PReLU(Conv2d(C,C,1)(Conv2d(C,C,(3,3),pad (1,d),dil (1,d),groups=C)(x (+ x2)))))
 * replacing, per layer, the depthwise conv, the pointwise conv, the shared PReLU, the BatchNorm2d
 * (eval mode, folded to bn_scale/bn_shift) and the running-value add of UNet_m.py:545-561 -- five tensor
 * passes -- by one read and one write.  residual = 0 for the block's first layer (r = L0(x)), 1 for
 * the others (r = r + Li(r)); x2 (or null) is the skip tensor torch.add'ed in front of a decoder's
 * block (ResUNet.py:209).  dilation_w in {1,2,4,8,16}.  dw_weight [C,1,3,3], pw_weight [C,C,1,1] (torch
 * layouts), dw_bias / pw_bias [C] or null, prelu_alpha [1], bn_scale / bn_shift [C]; out must not alias
 * x or x2; x2 and residual are not combined (the reference never does).  Returns UNETMI_E_UNSUPPORTED for other channel counts (use the unfused entry points). */
int unetmi_resblock_layer_fwd(int N, int C, int H, int W, int dilation_w, const void* x, const void* x2,
                              const float* dw_weight, const float* dw_bias, const float* pw_weight,
                              const float* pw_bias, const float* prelu_alpha, const float* bn_scale,
                              const float* bn_shift, int residual, void* out, int dtype, void* stream);

/* Pointwise (1x1, stride 1) convolution 64 -> 64 channels with the fused epilogue, as one bandwidth-bound
 * pass: out = epilogue(weight[64,64] . (x (+ x2)) + bias) (+ x when residual).  The epilogue is
 * act(v * scale + shift) when act_first = 0 (conv -> BN -> act: the 1x1 res_paths of ResUNet,
 * ResUNet.py:125-132 through UNet_m.py:340-345) or act(v) * scale + shift when act_first = 1; act is
 * LeakyReLU / single-parameter PReLU with slope leaky_slope[0], or the identity when leaky_slope is null.
 * Same tensors as unetmi_conv2d_fwd would take for that layer, except weight in torch layout [64,64,1,1]. */
int unetmi_pwconv64_fwd(int N, int H, int W, const void* x, const void* x2, const float* weight, const float* bias,
                        const float* scale, const float* shift, const float* leaky_slope, int act_first, int residual,
                        void* out, int dtype, void* stream);

/* Time recurrence of one layer of a projected LSTM (nn.LSTM(..., hidden 640, proj_size P), the bottleneck of
 * ResUNet.py:139,186-197) as one persistent launch.  The caller does the time-parallel parts as GEMMs:
 *   gx[t][g*H+u][b]  = (W_ih x_t + b_ih + b_hh)[g*H+u]  for all t     (gate order i, f, g, o)
 *   wrec[4H][H]      = W_hh W_hr                                       (recurrent and projection matrix folded)
 * and reads the pre-projection states back:  hfull[t+1][u][b] = o_t * tanh(c_t);  h_t = W_hr hfull[t+1].
 * hfull is [T+1][HP][BP], HP = H rounded up to a multiple of 16, with slot 0 = the initial state (zeros for the
 * reference's call) and rows H..HP-1 of every slot zero (the caller clears the buffer once; the kernel never
 * writes them); BP = batch padded to a multiple of 16 (<= 64; padded columns compute on zeros and are ignored).
 * sync_ws: 4 * (H/4 + 1) bytes of device scratch (cleared by the call).  H must be 640 (ResUNet.py:139) or 300
 * (ResUNet.py:335, UNet_m.py:587-610).  All workgroups have to be resident at once (checked with
 * the runtime's occupancy query); the call refuses otherwise.  If the device-wide barrier inside ever times out
 * (a workgroup that is not resident after all, e.g. the GPU is shared), every slot of hfull from the step that
 * timed out on is filled with NaN and the last word of sync_ws, ((uint32_t*)sync_ws)[H/4], is left non-zero
 * when the launch completes -- the status a caller can fold into its result without a host round trip. */
int unetmi_lstm_recurrence(int T, int BP, int H, const float* wrec, const float* gx, float* hfull, void* sync_ws,
                           void* stream);

/* The recurrence of a single-layer, unidirectional nn.GRU (DSUNet.py:281-312 TGRUBlock: GRU(128,128); UDSS.py's
 * TGRUBlock GRU(256,512) and cRNN GRU(256,256)) as one persistent launch.  The caller supplies
 *   gx[t][g*H+u][b] = (W_ih x_t + b_ih)[g*H+u] for all t   (gate order r, z, n: torch's weight_ih_l0 / bias_ih_l0),
 *   laid out [T][3H][BP], or with gx_channel_major != 0 as [3H][T][BP] (what a 1x1 convolution over a
 *   [channels, T, BP] image writes)
 *   whh[3H][H] = weight_hh_l0, bhh[3H] = bias_hh_l0
 * and hfull [T+1][H][BP] with slot 0 = the initial state; the kernel fills slots 1..T with h_1..h_T (the GRU's
 * outputs).  BP = batch padded to a multiple of 16 (<= 64); sync_ws: 4 * (H/4 + 1) bytes of device scratch
 * (cleared by the call), whose last word is non-zero after an aborted run (all state rows from the aborted step on
 * are then NaN).  H must be 128, 256 or 512; all H/4 workgroups have to be resident at once (checked). */
int unetmi_gru_recurrence(int T, int BP, int H, const float* whh, const float* bhh, const float* gx,
                          int gx_channel_major, float* hfull, void* sync_ws, void* stream);

/* One time step of a GRU for M independent sequences at once -- the elementwise half of the bidirectional
 * frequency-axis GRU (UDSS.py:15-37, UNet_m.py:615-638: nn.GRU(..., bidirectional=True) over B*T sequences of a few
 * steps), whose recurrent half is then one library GEMM per step (gh = h W_hh^T):
 *   r = sigmoid(gx_r + gh_r + b_hr), z = sigmoid(gx_z + gh_z + b_hz), n = tanh(gx_n + r (gh_n + b_hn)), h' = (1-z) n + z h.
 * gx: row m at gx + m * gx_stride, 3H values (W_ih x + b_ih, gates r, z, n); gh [M][3H] contiguous or null (first step
 * from a zero state: no product needed); bhh [3H]; hprev: row m at hprev + m * hprev_stride, or null (zero state);
 * out: row m at out + m * out_stride, H values.  Strides in elements. */
int unetmi_gru_gate_fwd(long long M, int H, const float* gx, long long gx_stride, const float* gh, const float* bhh,
                        const float* hprev, long long hprev_stride, float* out, long long out_stride, void* stream);

/* The GRU under autograd (train() mode of the TGRU bottlenecks and of cRNN, DSUNet.py:281-312, UDSS.py:39-70, :190-239;
 * the reference leaves the backward to torch autograd).  Step-major layouts, sizes and sync_ws as unetmi_gru_recurrence.
 *
 * unetmi_gru_recurrence_train: the forward recurrence that also fills gates [T][4H][BP] with r, z, q = W_hn h + b_hn
 *   and n of every step (rows g*H + u in that order).
 *
 * unetmi_gru_bptt: the reverse recurrence as ONE persistent launch.  dy [T][H][BP] = gradient of the loss w.r.t. the
 *   outputs h_1..h_T (the gradient of the final state, if any, added to the last step by the caller).  gates is
 *   overwritten in place: rows [0, 3H) become the gradient w.r.t. W_hh h + b_hh (gates r, z, n), rows [0, 2H) and
 *   [3H, 4H) the gradient w.r.t. W_ih x + b_ih, so that  dW_hh = sum_t G_t[0:3H] h_{t-1}^T,
 *   dW_ih = sum_t G_t[0:2H, 3H:4H] x_t^T  and  dx_t = W_ih^T G_t[0:2H, 3H:4H]  are time-batched products afterwards;
 *   dh0 [H][BP] receives the gradient reaching the initial state.  H/16 workgroups, all resident at once (checked);
 *   after an aborted run (last word of sync_ws non-zero) every gradient from that step down is NaN. */
int unetmi_gru_recurrence_train(int T, int BP, int H, const float* whh, const float* bhh, const float* gx,
                                float* hfull, float* gates, void* sync_ws, void* stream);
int unetmi_gru_bptt(int T, int BP, int H, const float* whh, float* gates, const float* hfull, const float* dy,
                    float* dh0, void* sync_ws, void* stream);

/* The same recurrence with the batch split over the chip's 8 XCDs (4 batch columns each, BP <= 32): every XCD runs
 * an independent recurrence for its columns, the state exchange stays inside that XCD's L2 and there is no
 * device-wide barrier (about 2 us per step instead of 6-7).  Arguments and results as unetmi_lstm_recurrence.
 * ws: unetmi_lstm_recurrence_xcd_workspace_bytes() bytes of device scratch (cleared by the call).  H = 640 (one
 * workgroup on each of the 256 compute units) or 300 (padded to 320 inside: 16 workgroups per XCD).  Needs a
 * device of 8 XCDs x 32 compute units with one workgroup resident per compute unit (UNETMI_E_UNSUPPORTED
 * otherwise: use unetmi_lstm_recurrence).  A workgroup finds its XCD from the hardware, not from its index.
 * Status: ((uint32_t*)ws)[8] is non-zero after an aborted run (a workgroup was not resident, e.g. the GPU is
 * shared); every slot of hfull from the aborted step on is then NaN.
 * channel_major = 1 takes gx as [4H][T+1][BP] (slot t+1 = step t; slot 0 is not read) and writes hfull as
 * [H][T+1][BP]: the layouts in which the two matrix products around the recurrence are 1x1 convolutions of
 * unetmi_conv2d_fwd over a [channels, T+1, BP] image (gx = W_ih * x + b; h = W_hr * hfull), so a layer's output
 * is the next layer's input without a transpose.  channel_major is a flags word: bit 0 as above, bit 1 (value 2)
 * = the recurrent product of each step with fp16 operands (wrec and the previous state rounded to fp16 for the matrix
 * instruction, a quarter of the step's matrix work; accumulation, gates, cell and state stay fp32), for callers
 * whose sequence was rounded to 16 bits anyway. */
size_t unetmi_lstm_recurrence_xcd_workspace_bytes(void);
int unetmi_lstm_recurrence_xcd(int T, int BP, int H, const float* wrec, const float* gx, float* hfull,
                               int channel_major, void* ws, size_t ws_bytes, void* stream);

/* The projected LSTM under autograd (train() mode of ResUNet: nn.LSTM(384, 640, 3, proj_size=384), ResUNet.py:139,
 * :186-197, whose backward the reference leaves to torch autograd; also the hidden-size-300 LSTMs of ResUNet.py:335 and
 * UNet_m.py:587-610).  Channel-major layouts only, H = 640 or 300, BP <= 32,
 * same device requirements, workspace rules and status word as unetmi_lstm_recurrence_xcd.
 *
 * unetmi_lstm_recurrence_xcd_train: the forward recurrence that also keeps what backpropagation through time needs.
 *   gx_gates [4H][T+1][BP]: in = input contributions (slot t+1 = step t), out = the ACTIVATED gates i, f, g, o of
 *   every step in the same places; hfull [H][T+1][BP] as above; cell [H][T+1][BP]: c_t at slot t+1 (the caller
 *   zeroes slot 0 of hfull and cell).  ws: unetmi_lstm_recurrence_xcd_workspace_bytes().
 *
 * unetmi_lstm_bptt_xcd: the reverse recurrence.  With m_t = o_t tanh(c_t) and Wrec = W_hh W_hr,
 *   dm_t = dhfull_t + Wrec^T G_{t+1},  dc_t = dm_t o_t (1 - tanh^2 c_t) + dc_{t+1} f_{t+1},
 *   G_t = gradient of the loss w.r.t. step t's four gate pre-activations,
 * for t = T-1 .. 0 as ONE persistent launch (G is what the XCD's workgroups exchange per step).
 *   gates [4H][T+1][BP]: in = what the forward call left, out = G_t at slot t+1 (slot 0 untouched);
 *   cell as written by the forward call; dhfull [H][T+1][BP] = W_hr^T dL/dy_t at slot t+1 (one matrix product by
 *   the caller).  The weight gradients are time-batched matrix products over G afterwards:
 *   dW_ih = sum_t G_t x_t^T, db = sum_t G_t, dWrec = sum_t G_t m_{t-1}^T, dx_t = W_ih^T G_t.
 *   ws: unetmi_lstm_bptt_xcd_workspace_bytes() bytes of device scratch (cleared by the call); status word
 *   ((uint32_t*)ws)[8]; after an aborted run every G from that step down is NaN. */
int unetmi_lstm_recurrence_xcd_train(int T, int BP, int H, const float* wrec, float* gx_gates, float* hfull,
                                     float* cell, void* ws, size_t ws_bytes, void* stream);
size_t unetmi_lstm_bptt_xcd_workspace_bytes(void);
int unetmi_lstm_bptt_xcd(int T, int BP, int H, const float* wrec, float* gates, const float* cell,
                         const float* dhfull, void* ws, size_t ws_bytes, void* stream);

/* LayerNorm over the FREQUENCY axis of x[N,C,F,T] (biased variance, affine gamma/beta[F]),
 * written into channels [c_off, c_off+C) of out[N,C_out,F,T].  Replaces
 *   permute -> nn.LayerNorm(257) on [B,C,T,F] -> permute -> torch.cat       ResUNet.py:165-175, :49, :53
 * with the two permutes folded into the indexing and the cat into the channel offset. */
int unetmi_layernorm_f_fwd(int N, int C, int F, int T, int C_out, int c_off, float eps, const void* x,
                           const float* gamma, const float* beta, void* out, int dtype, void* stream);

/* out[n,k,f,t] = (tanh(y[n,k,f,t]) + eps) * x[n,k,f,t] with x holding Cx >= K channels of which
 * the first K are used.  Replaces  mask = tanh(y) + 1e-13 ; output = mask * input[:, :2]
 *   ResUNet.py:217-219.   y, out: [N,K,F,T]; x: [N,Cx,F,T]; device fp32. */
int unetmi_tanh_mask_mul(int N, int K, int Cx, int F, int T, float eps, const void* y, const void* x, void* out,
                         int dtype, void* stream);

/* InstanceNorm2d + activation of an Encoder / Decoder built with norm="InstanceNorm2d" (UNet_m.py:312-345,
 * DSUNet.py:62-93 with type_norm="InstanceNorm2d"): out = gate[n,c] * act(gamma[c] * (z - mean[n,c]) * rstd[n,c]
 * + beta[c]) with the statistics of each (n, c) plane over its HW values (biased variance, eps inside the root),
 * which nn.InstanceNorm2d uses in train() and eval() alike unless it tracks running statistics.  gamma / beta /
 * gate may be null (InstanceNorm2d's default has no affine parameters); act_param: the slope of LeakyReLU / PReLU,
 * one value or one per channel.  One launch, the plane is streamed twice. */
int unetmi_instnorm_act_fwd(int N, int C, int HW, const void* z, const float* gamma, const float* beta, float eps, int act,
                            const float* act_param, int act_param_per_channel, const float* gate, void* out, int dtype,
                            void* stream);

/* LayerNorm + activation of an Encoder / Decoder built with norm="LayerNorm" (UNet_m.py:312-319, :355-362:
 * nn.LayerNorm(out_channels) applied to the NCHW convolution output normalises each row (n, c, h) over its W values,
 * so W must equal out_channels): out = gate[n,c] * act(gamma[w] * (z - mean_row) * rstd_row + beta[w]), biased
 * variance, eps inside the root.  gamma / beta [W] and gate [N*C] may be null; act_param: one slope or one per
 * channel.  One launch, one wave per row. */
int unetmi_layernorm_w_act_fwd(int N, int C, int H, int W, const void* z, const float* gamma, const float* beta, float eps,
                               int act, const float* act_param, int act_param_per_channel, const float* gate, void* out,
                               int dtype, void* stream);

/* Gate stage of DSUNet's gated convolutions (GConv / TrGConv, DSUNet.py:134-172, :241-275), which compute
 * sigmoid(conv_m(x)) * conv_c(x) [-> norm -> activation].  The two convolutions share geometry and input, so
 * they run as ONE unetmi_conv2d_fwd launch with the weights stacked to 2K output channels (features first,
 * mask logits second); this entry point finishes the layer in one bandwidth-bound pass:
 *   out[n,k,i] = gate[n,k] * act(scale[k] * (sigmoid(z[n,K+k,i]) * z[n,k,i]) + shift[k])
 * z [N,2K,HW], out [N,K,HW] (UNETMI_DTYPE_* storage); scale/shift (folded eval BatchNorm) [K] or null,
 * act_param [1] or null, gate [N,K] or null (DSUNet's attractor gate, DSUNet.py:599-602). */
int unetmi_glu_fwd(int N, int K, int HW, const void* z, const float* scale, const float* shift, int act,
                   const float* act_param, const float* gate, void* out, int dtype, void* stream);

/* ---- bfloat16-operand convolution on a channel-blocked layout ---------------------------------
 * The same layers as unetmi_conv2d_fwd (Encoder / Decoder forward with the concat, eval BatchNorm,
 * activation and DSUNet's gate fused: UNet_m.py:340-345, :379-383, DSUNet.py:86-93, :194-200, :599-602, :622)
 * for activations kept between layers as  [N][CB][H][W][8] bfloat16,  CB = 2*ceil(C/16)  (eight
 * consecutive channels of a pixel form one 16-byte record; channels padded with zeros to a multiple of 16).
 * Products run on the bf16 matrix instruction with fp32 accumulation; bias / BatchNorm / activation /
 * gate are applied to the fp32 accumulator.  Descriptor as for unetmi_conv2d_fwd, with
 *   in_mode      UNETMI_IN_SINGLE or UNETMI_IN_CONCAT (no add, no residual); K <= 128, or a multiple of 128 (the
 *                launch then has one grid column per 128-channel chunk and the weight image one image per chunk),
 *   reserved[1]  weight halves: 0/1 = weights rounded to bfloat16; 2 = a second image holds the rounding
 *                remainder and a second MFMA adds it (fp32-weight accuracy at twice the matrix work),
 *   reserved[2]  bits: UNETMI_B16_OUT_NCHW_F32 = write out as plain fp32 NCHW [N,K,OH,OW] instead of the blocked
 *                layout (heads); UNETMI_B16_IN_F16 = both inputs and the weight image hold fp16 instead of bfloat16
 *                (same layout, same matrix rate, three more mantissa bits; a layer's inputs share one format);
 *                UNETMI_B16_OUT_F16 = the blocked output is stored as fp16 (a value beyond +-65504 becomes infinity and
 *                shows as NaN downstream -- it is not clipped).
 *                fp16 suits tensors of known modest range -- activations behind a BatchNorm -- and cuts the
 *                rounding error of a chain of layers about eightfold; a network input of unknown scale stays
 *                bfloat16. */
enum { UNETMI_B16_OUT_NCHW_F32 = 1, UNETMI_B16_IN_F16 = 2, UNETMI_B16_OUT_F16 = 4 };
size_t unetmi_conv2d_b16_packed_weight_bytes(const UnetmiConvDesc* d);
/* HOST function, a measurement aid like unetmi_conv2d_kernel_name: the kernel instantiation unetmi_conv2d_b16_fwd
 * (head_channels = 0) or unetmi_conv2d_b16_head_fwd launches for this descriptor, e.g. "conv_b16_kernel_rt2<3>". */
int unetmi_conv2d_b16_kernel_name(const UnetmiConvDesc* d, int head_channels, char* buf, size_t n);
/* HOST function; w: torch-layout fp32 weight, packed: host buffer of the size above. */
int unetmi_conv2d_b16_pack_weights(const UnetmiConvDesc* d, const float* w, void* packed);
int unetmi_conv2d_b16_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                          const float* bias, const float* scale, const float* shift, const float* act_param,
                          const float* gate, void* out, void* stream);
/* unetmi_conv2d_b16_fwd writing plain fp32 NCHW [N,K,OH,OW] (UNETMI_B16_OUT_NCHW_F32 is taken as set) with an
 * optional fp32 NCHW tensor of that shape added to the result -- the `y + res` in front of the next block of an
 * additive-skip decoder (ResUNet.py:206-210).  Together with unetmi_blocked_from_nchw_sum it lets a network whose
 * tensors stay fp32 NCHW run single convolutions with bfloat16 operands (ResUNet's strided Encoder / Decoder
 * blocks on a bfloat16 input, ResUNet.py:59-113). */
int unetmi_conv2d_b16_nchw_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                               const float* bias, const float* scale, const float* shift, const float* act_param,
                               const float* gate, const float* residual, float* out, void* stream);
/* The last decoder of such a chain together with the network's 1x1 mask head and its squash
 * (DSUNet.py:616-628 with CRM.forward :355-359: decoder -> Conv2d(C, 2, 1) -> tanh / sigmoid / softplus): the
 * decoder's result stays in the fp32 accumulators (it is never rounded or written), and per pixel
 *   out[p] = head_act(sum_k head_w[p][k] * decoder_out[k] + head_b[p]),   p < head_channels <= 4.
 * head_w [head_channels][K] fp32, head_b [head_channels] or null, head_act = UNETMI_ACT_* (not LeakyReLU, not GELU).
 * out: fp32 [N][head_channels][OH][OW], or with interleaved != 0 [N][OH][OW][head_channels] (two channels that
 * way are a complex64 [N][OH][OW] tensor, which is what the reference's CRM head returns).  Of the descriptor's
 * reserved[2] only UNETMI_B16_IN_F16 matters.  Not built for time-stride-2 transposed layers (UNETMI_E_BAD_DESC). */
int unetmi_conv2d_b16_head_fwd(const UnetmiConvDesc* d, const void* in0, const void* in1, const void* wpacked,
                               const float* bias, const float* scale, const float* shift, const float* act_param,
                               const float* gate, const float* head_w, const float* head_b, int head_channels,
                               int head_act, int interleaved, float* out, void* stream);
/* Layout conversion either side of such a chain: src NCHW [N,C,HW] (UNETMI_DTYPE_F32 / _BF16) -> blocked bfloat16,
 * and back.  These have no counterpart in the reference (its tensors are NCHW throughout). */
int unetmi_blocked_from_nchw(int N, int C, int HW, const void* src, int src_dtype, void* dst, void* stream);
/* The same with a second tensor of the same shape and storage type added first (src + src2 in fp32, rounded once;
 * src2 may be null): the summed skip in front of a decoder (ResUNet.py:209) costs no pass of its own.
 * dst_dtype: UNETMI_DTYPE_BF16 or UNETMI_DTYPE_F16, the 16-bit format of the blocked tensor. */
int unetmi_blocked_from_nchw_sum(int N, int C, int HW, const void* src, const void* src2, int src_dtype, void* dst,
                                 int dst_dtype, void* stream);
/* src_dtype: UNETMI_DTYPE_BF16 or UNETMI_DTYPE_F16 (the blocked tensor); dst_dtype: UNETMI_DTYPE_F32 or _BF16 (NCHW). */
int unetmi_nchw_from_blocked(int N, int C, int HW, const void* src, int src_dtype, void* dst, int dst_dtype, void* stream);


/* Complex ratio filter / complex mask application over the STFT:
 *   Y[b,n,c,f,t] = sum_{p<PF, q<PT} Hf[b,n,c,p,q,f,t] * X[b,c,f+p-(PF-1)/2, t+q-(PT-1)/2]   (zero outside)
 * Hf: complex64 [B,N,C,PF,PT,F,T] (the tensor cRFUNet.UNet20.forward returns, cRFUNet.py:320-324);
 * X: complex64 [B,C,F,T]; Y: complex64 [B,N,C,F,T]; all passed as interleaved float pairs.
 * With N = C = PF = PT = 1 it is the complex mask product X * M of CRM.output (DSUNet.py:361-362).
 * The reference only EMITS the ratio filter; its application has no counterpart there, so this
 * definition is this library's own (checked against its CPU restatement only). */
int unetmi_crf_apply(int B, int N, int C, int PF, int PT, int F, int T, const float* Hf, const float* X, float* Y,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (train()-mode forward and backward of Encoder / Decoder, UNet_m.py:340-345,
 * :379-383, under torch autograd in the reference).  Train-mode forward of a block is
 *   z = unetmi_conv2d_fwd(act NONE, scale/shift null)          conv / transconv (+ bias)
 *   unetmi_bn_train_stats(z)  -> batch mean, 1/sqrt(var+eps), running-stat update
 *   y = unetmi_bn_act_fwd(z)                                   bn -> acti
 * and its backward
 *   dz, dgamma, dbeta, dbias = unetmi_bn_act_bwd(dy, z)
 *   dW = unetmi_conv2d_wgrad(input, dz)
 *   dx = unetmi_conv2d_fwd with the transposed flag flipped (a conv's input gradient is the
 *        transposed conv of dz with the same weight tensor, and vice versa).
 * Workspaces are caller-allocated device memory; their contents are scratch.
 * ------------------------------------------------------------------------------------------ */

/* Bytes of workspace unetmi_bn_train_stats / unetmi_bn_act_bwd need for z[N,C,HW]. */
size_t unetmi_bn_workspace_bytes(int N, int C, int HW);

/* Batch statistics of z[N,C,HW] over (N,HW) per channel (nn.BatchNorm2d in train mode, UNet_m.py:322):
 * mean[c], invstd[c] = 1/sqrt(biased_var + eps); when non-null, running_mean / running_var are
 * updated in place as r = (1-momentum) r + momentum * stat with the UNBIASED variance. */
int unetmi_bn_train_stats(int N, int C, int HW, const float* z, float eps, float momentum, float* running_mean,
                          float* running_var, float* mean, float* invstd, void* ws, size_t ws_bytes, void* stream);

/* unetmi_bn_train_stats for callers that cannot allocate device memory or bring a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * Same arguments and results as unetmi_bn_train_stats without ws / ws_bytes: the per-block partial sums go to a
 * 256 KiB array in the library's own device image (one copy per GPU, held until the process ends).  Limit:
 * C * splits * 16 bytes <= 256 KiB, where splits is the number of blocks a channel's N*HW values are divided into
 * (about 4096 / C, at most 1024): every layer of the reference's networks fits (C <= 384 at any size); larger
 * requests return UNETMI_E_UNSUPPORTED.  Calls on the same GPU share the array: two calls that overlap in time on
 * one GPU (different streams or threads) race; calls on one stream are ordered and safe.  Calls no hipMalloc,
 * hipFree or any allocating routine.  `stream` must belong to the calling thread's current device (checked:
 * UNETMI_E_BAD_ARG otherwise).  The FIRST call on a device looks the array's address up (hipGetSymbolAddress,
 * which may load the code object): make that one call outside graph capture; every later call on the device
 * only launches two kernels. */
int unetmi_bn_train_stats_static(int N, int C, int HW, const float* z, float eps, float momentum,
                                 float* running_mean, float* running_var, float* mean, float* invstd, void* stream);

/* out = act((z - mean) * invstd * gamma + beta), act = UNETMI_ACT_* with one shared parameter
 * (act_param[0], LeakyReLU slope) or none. */
int unetmi_bn_act_fwd(int N, int C, int HW, const float* z, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int act, const float* act_param, float* out,
                      void* stream);

/* Backward of unetmi_bn_act_fwd composed with train-mode batch statistics:
 *   g = dy * act'(u),  dbeta = sum g,  dgamma = sum g * xhat,
 *   dz = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),  dbias = sum dz  (the conv bias gradient).
 * dy is [N,C,HW] with element (n,c,i) at dy[n * dy_batch_stride + c * HW + i], so a channel slice
 * of a wider gradient tensor can be passed without a copy (dy_batch_stride = C*HW when dense).
 * With batch_stats = 0 mean/invstd are constants (eval-mode BatchNorm, or mean 0 / invstd 1 / gamma 1 /
 * beta 0 for a layer without a norm): dz = gamma * invstd * g, the two mean terms drop out.
 * dgamma / dbeta / dbias may be null. */
int unetmi_bn_act_bwd(int N, int C, int HW, const float* dy, long long dy_batch_stride, const float* z,
                      const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                      const float* act_param, int batch_stats, float* dz, float* dgamma, float* dbeta,
                      float* dbias, void* ws, size_t ws_bytes, void* stream);

/* Weight gradient of the layer d describes (same descriptor as the forward; epilogue fields ignored).
 *   transposed = 0:  dweight[K][C0][kh][kw]      = correlate(dout[N,K,OH,OW], in0[N,C0,H,W])
 *   transposed = 1:  dweight[C0+C1][K][kh][kw]   = correlate(cat(in0,in1)[N,C0+C1,H,W], dout[N,K,OH,OW])
 * i.e. what autograd computes for nn.Conv2d.weight / nn.ConvTranspose2d.weight (UNet_m.py:321, :361).
 * Deterministic (fixed summation order).  ws: unetmi_conv2d_wgrad_workspace_bytes(d) bytes. */
size_t unetmi_conv2d_wgrad_workspace_bytes(const UnetmiConvDesc* d);
int unetmi_conv2d_wgrad(const UnetmiConvDesc* d, const float* in0, const float* in1, const float* dout,
                        float* dweight, void* ws, size_t ws_bytes, void* stream);

/* One Adam update (torch.optim.Adam defaults: no amsgrad; weight_decay is L2 added to the gradient) over
 * a flat bucket of n parameters: g = grad * grad_scale (+ weight_decay * p); m, v updated in place;
 * p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps).  step counts from 1.
 * grad_scale = 1 / world_size turns a summed all-reduce into the data-parallel mean. */
int unetmi_adam_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                     float beta1, float beta2, float eps, float weight_decay, float grad_scale, int step,
                     void* stream);

/* Train-mode BatchNorm2d's running-statistics update (UNet_m.py:312 under train()) from the (mean, invstd) that
 * unetmi_bn_train_stats returned for a batch of `count` values per channel:
 *   running_mean = (1 - momentum) running_mean + momentum mean
 *   running_var  = (1 - momentum) running_var  + momentum (1/invstd^2 - eps) count / (count - 1).
 * For callers that keep the statistics kernel free of side effects (pass it null running pointers). */
int unetmi_bn_running_update(int C, long long count, float eps, float momentum, const float* mean, const float* invstd,
                             float* running_mean, float* running_var, void* stream);

/* dst[i] = map[i] >= 0 ? src[map[i]] : 0 for i < n.  With the map of a layer (the element numbers that
 * unetmi_conv2d_pack_weights puts at each position of the packed image, -1 for padding; found once per layer
 * shape by packing a numbered tensor on the host) this re-packs a weight tensor on the device after every
 * optimizer step.  No counterpart in the reference, whose convolutions read nn.Conv2d.weight directly. */
int unetmi_gather_f32(size_t n, const float* src, const int* map, float* dst, void* stream);

/* The training step's gradient exchange: an in-place fp32 SUM all-reduce over RCCL, one process per GPU, one
 * communicator per process.  The reference has no distributed code (its modules are trained by user scripts);
 * this replaces what such a script would do with torch.distributed.all_reduce on the gradients of the modules
 * of UNet.py:7-113.  The RCCL library is located when the communicator is created (the copy already loaded in
 * the process, else the system's); without one these calls return UNETMI_E_UNSUPPORTED and every other entry
 * point still works.
 *   unetmi_comm_unique_id  rank 0 fills UNETMI_COMM_ID_BYTES opaque bytes and hands them to the other ranks
 *                          through the job's own channel (TCP store, file, MPI)
 *   unetmi_comm_init       every rank, with the GPU it owns current; collective (returns when all have joined)
 *   unetmi_allreduce_f32   enqueues buf[i] = sum over ranks of buf[i] on `stream`; divide by the world size in
 *                          unetmi_adam_step's grad_scale
 *   unetmi_comm_world      the communicator's size, 0 without one
 *   unetmi_comm_destroy    releases it (a later unetmi_comm_init may create another) */
#define UNETMI_COMM_ID_BYTES 128
int unetmi_comm_unique_id(void* id128);
int unetmi_comm_init(int rank, int world, const void* id128);
int unetmi_comm_world(void);
int unetmi_allreduce_f32(float* buf, size_t n, void* stream);
int unetmi_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* UNETMI_H */
