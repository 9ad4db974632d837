"""Projected LSTM on the device (the bottleneck of ResUNet, ResUNet.py:139,186-197).

Everything without a time dependency is a large matrix product: the input contribution of all time
steps before the recurrence, the projection of all pre-projection states after it.  With the XCD
kernel they are 1x1 convolutions of the library's own kernel on a [channels, T+1, BP] layout that
the recurrence kernel reads and writes directly (``GEMM = "conv"``); the device-wide-barrier kernel
keeps the step-major layout and torch.matmul.  The recurrence itself is one persistent HIP launch per
layer (unetmi_lstm_recurrence_xcd / unetmi_lstm_recurrence) instead of three launches per time step.

Under autograd (``lstm_forward_train``; train() mode of ResUNet) the forward recurrence also keeps the activated
gates and the cell states, and backpropagation through time is ONE persistent launch per layer as well
(unetmi_lstm_bptt_xcd) between time-batched matrix products for the weight and input gradients.
"""

import torch
import torch.nn as nn

from . import ops as _ops

MAX_BATCH_PER_LAUNCH = 64          # device-wide-barrier kernel (unetmi_lstm_recurrence)
XCD_BATCH_PER_LAUNCH = 32          # XCD-local kernel (unetmi_lstm_recurrence_xcd): 8 XCDs x 4 batch columns

# Which recurrence kernel lstm_forward uses: "xcd" (batch split over the XCDs, state exchange inside each XCD's
# L2; falls back to "grid" on a device it does not fit) or "grid" (every workgroup reads the whole state, one
# device-wide barrier per step).
KERNEL = "xcd"

# What lstm_forward does when the XCD kernel reports that a wait gave up (its 32 workgroups per XCD must all be
# resident at once; a GPU shared with other work can starve one): "rerun" (default) reads the status word on the
# host -- one synchronisation per call -- and computes the call again with the device-wide-barrier kernel, whose
# host side checks residency before launching; "nan" keeps everything asynchronous and turns the output into NaN.
ON_ABORT = "rerun"


def supported(lstm):
    return (isinstance(lstm, nn.LSTM) and lstm.hidden_size in (640, 300) and lstm.proj_size > 0 and lstm.batch_first
            and not lstm.bidirectional and lstm.bias and (lstm.dropout == 0.0 or not lstm.training))


def _folded_recurrent_matrix(lstm, layer, cache):
    """W_hh W_hr in float64, rounded once to fp32; cached on the parameters' versions."""
    w_hh = getattr(lstm, "weight_hh_l{}".format(layer))
    w_hr = getattr(lstm, "weight_hr_l{}".format(layer))
    key = (w_hh._version, w_hh.data_ptr(), w_hr._version, w_hr.data_ptr())
    hit = cache.get(layer)
    if hit is None or hit[0] != key:
        with torch.no_grad():
            wrec = (w_hh.detach().double() @ w_hr.detach().double()).float().contiguous()
        cache[layer] = hit = (key, wrec)
    return hit[1]


def _xcd_kernel_fits(dev, hidden):
    return KERNEL == "xcd" and hidden in (640, 300) and torch.cuda.get_device_properties(dev).multi_processor_count == 256


def _recurrence(wrec, gx, hfull, use_xcd, channel_major=False, f16=False):
    """One layer's T dependent steps (torch.ops.unetmi.lstm_recurrence); returns the kernel's status word
    (a device scalar, non-zero = aborted)."""
    return _ops.lstm_recurrence(wrec, gx, hfull, bool(use_xcd), bool(channel_major), bool(f16))


# With products = "fp16" the recurrent matrix-vector product of every step takes fp16 operands as well (fp32
# accumulation, gates, cell and state): a quarter of the step's matrix instructions.  False keeps that product fp32.
FP16_RECURRENCE = True


# The two matrix products around the recurrence -- the input contributions of all steps, and the projection of all
# pre-projection states -- as 1x1 convolutions of the library's own kernel over a [channels, T+1, BP] image
# ("conv"), or as torch.matmul, i.e. the ROCm BLAS library ("blas": the step-major layout; what the device-wide-
# barrier kernel uses).
GEMM = "conv"
_pointwise_cache = {}


def _pointwise(weight, bias, cache, key):
    """FusedConv of a [K, C] matrix as a 1x1 convolution (+ bias), cached on the parameters' versions."""
    from .fused import FusedConv
    ver = (weight._version, weight.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()))
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        K, C = weight.shape
        conv = nn.Conv2d(C, K, 1, bias=bias is not None).to(weight.device)
        with torch.no_grad():
            conv.weight.copy_(weight.detach().reshape(K, C, 1, 1))
            if bias is not None:
                conv.bias.copy_(bias.detach())
        cache[key] = hit = (ver, FusedConv(conv))
    return hit[1]


def _pointwise16(weight, bias, cache, key):
    """The same matrix as a 1x1 convolution with fp16 operands (unet_amd/blocked.py: fp32 accumulation, fp32
    channel-major output), cached on the parameters' versions."""
    from .blocked import BlockedConv
    ver = (weight._version, weight.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()))
    hit = cache.get(key)
    if hit is None or hit[0] != ver:
        K, C = weight.shape
        conv = nn.Conv2d(C, K, 1, bias=True).to(weight.device)
        with torch.no_grad():
            conv.weight.copy_(weight.detach().reshape(K, C, 1, 1))
            conv.bias.copy_(bias.detach()) if bias is not None else conv.bias.zero_()
        cache[key] = hit = (ver, BlockedConv(conv, None, None, nchw_out=True))
    layer = hit[1]

    def run(image):                                   # [1, C, T+1, BP] fp32 -> [1, K, T+1, BP] fp32
        from .blocked import BlockedTensor
        return layer(BlockedTensor.from_nchw(image, fmt="fp16"))
    return run


def _forward_channel_major(lstm, x, cache, products="fp32"):
    """x [B, T, input] -> [B, T, proj] with every tensor of the layer stack kept as [channels, T+1, BP] (slot 0 =
    the time before the first step): the products are 1x1 convolutions, a layer's projected output is the next
    layer's input as it stands, and the recurrence kernel reads and writes that layout directly.  ``products``:
    "fp32" (the exact-fp32 convolution kernel) or "fp16" (the two time-batched products with fp16 operands on the
    16-bit matrix instruction, for callers whose input was bfloat16 anyway; with FP16_RECURRENCE the recurrent product of
    each step takes fp16 operands too -- every sum, the gates, the cell and the state stay fp32)."""
    B, T, C = x.shape
    H = lstm.hidden_size
    BP = (B + 15) // 16 * 16
    dev = x.device
    seq = torch.zeros((1, C, T + 1, BP), dtype=torch.float32, device=dev)
    seq[0, :, 1:, :B] = x.permute(2, 1, 0)
    aborted = torch.zeros((), dtype=torch.int32, device=dev)
    for layer in range(lstm.num_layers):
        w_ih = getattr(lstm, "weight_ih_l{}".format(layer))
        w_hr = getattr(lstm, "weight_hr_l{}".format(layer))
        b_ih, b_hh = getattr(lstm, "bias_ih_l{}".format(layer)), getattr(lstm, "bias_hh_l{}".format(layer))
        bkey = ("bias", layer)
        bver = (b_ih._version, b_ih.data_ptr(), b_hh._version, b_hh.data_ptr())
        bhit = cache.get(bkey)
        if bhit is None or bhit[0] != bver:
            cache[bkey] = bhit = (bver, (b_ih.detach().float() + b_hh.detach().float()))
        pw = _pointwise16 if (products == "fp16" and w_ih.shape[0] % 128 == 0) else _pointwise
        gx = pw(w_ih, bhit[1], cache, ("ih", layer, products))(seq)               # [1, 4H, T+1, BP]
        wrec = _folded_recurrent_matrix(lstm, layer, cache)
        hfull = torch.empty((1, H, T + 1, BP), dtype=torch.float32, device=dev)
        hfull[0, :, 0].zero_()
        aborted = aborted | _recurrence(wrec, gx[0], hfull[0], True, channel_major=True,
                                        f16=products == "fp16" and FP16_RECURRENCE)
        pw = _pointwise16 if (products == "fp16" and (w_hr.shape[0] <= 128 or w_hr.shape[0] % 128 == 0)) else _pointwise
        seq = pw(w_hr, None, cache, ("hr", layer, products))(hfull)                # [1, proj, T+1, BP]
    out = seq[0, :, 1:, :B].permute(2, 1, 0).contiguous()
    return out, aborted


def lstm_forward(lstm, x, cache, _force_grid=False, products="fp32"):
    """x [B, T, input] -> [B, T, proj]; zero initial state, inference only.  ``products``: see
    _forward_channel_major."""
    if not supported(lstm):
        raise NotImplementedError("lstm_forward: built for batch_first projected LSTMs with hidden size 640 or 300")
    if not x.is_cuda or x.dtype != torch.float32:
        raise ValueError("lstm_forward: x must be an fp32 device tensor")
    B, T, _ = x.shape
    use_xcd = _xcd_kernel_fits(x.device, lstm.hidden_size) and not _force_grid
    group = XCD_BATCH_PER_LAUNCH if use_xcd else MAX_BATCH_PER_LAUNCH
    if B > group:                                   # sequences are independent: run them in groups
        return torch.cat([lstm_forward(lstm, x[i:i + group], cache, _force_grid, products) for i in range(0, B, group)], dim=0)
    if use_xcd and GEMM == "conv":
        with torch.no_grad(), torch.cuda.device(x.device):
            out, aborted = _forward_channel_major(lstm, x, cache, products)
            if ON_ABORT == "rerun":
                if int(aborted.item()) == 0:
                    return out
                import warnings
                warnings.warn("unet_amd.lstm: the XCD-local recurrence kernel gave up waiting for a workgroup that was "
                              "not resident (GPU shared with other work?); running the call again on the "
                              "device-wide-barrier kernel")
                return lstm_forward(lstm, x, cache, _force_grid=True, products=products)
            return torch.where(aborted != 0, torch.full_like(out, float("nan")), out)
    H = lstm.hidden_size
    BP = (B + 15) // 16 * 16
    dev = x.device
    with torch.no_grad(), torch.cuda.device(dev):
        seq = torch.zeros((T, x.shape[2], BP), dtype=torch.float32, device=dev)      # [T, in, BP]
        seq[:, :, :B] = x.permute(1, 2, 0)
        aborted = torch.zeros((), dtype=torch.int32, device=dev)
        for layer in range(lstm.num_layers):
            w_ih = getattr(lstm, "weight_ih_l{}".format(layer)).detach().float()
            w_hr = getattr(lstm, "weight_hr_l{}".format(layer)).detach().float()
            bias = (getattr(lstm, "bias_ih_l{}".format(layer)).float() + getattr(lstm, "bias_hh_l{}".format(layer)).float()).detach()
            gx = torch.matmul(w_ih, seq) + bias[None, :, None]                        # [T, 4H, BP]
            wrec = _folded_recurrent_matrix(lstm, layer, cache)
            HP = (H + 15) // 16 * 16                     # the barrier kernel's slots are padded to 16 rows (zeros)
            if HP == H:
                hfull = torch.empty((T + 1, H, BP), dtype=torch.float32, device=dev)
                hfull[0].zero_()
            else:
                hfull = torch.zeros((T + 1, HP, BP), dtype=torch.float32, device=dev)
            status = _recurrence(wrec, gx, hfull, use_xcd)
            aborted = aborted | status                                                # the kernel's status word, kept on the device
            seq = torch.matmul(w_hr, hfull[1:, :H])                                   # [T, proj, BP] = next layer's input
        out = seq[:, :, :B].permute(2, 0, 1).contiguous()
        # a recurrence that gave up waiting (GPU shared with other work) must not pass for a result: the whole
        # output becomes NaN, without a host round trip
        return torch.where(aborted != 0, torch.full_like(out, float("nan")), out)


# ------------------------------------------------------------------------------------------------------------
# train() mode: the same layer stack as an autograd node
# ------------------------------------------------------------------------------------------------------------
PARAMS_PER_LAYER = ("weight_ih_l{}", "weight_hh_l{}", "bias_ih_l{}", "bias_hh_l{}", "weight_hr_l{}")


def trainable_on_device(lstm, x):
    """lstm_forward_train serves this call: a projected LSTM of hidden size 640 or 300 (no dropout between the
    layers) on a device the XCD-local kernels fit; other devices keep torch's device LSTM under autograd."""
    return (supported(lstm) and x.is_cuda and x.dtype == torch.float32 and _xcd_kernel_fits(x.device, lstm.hidden_size))


class RecurrenceAborted(RuntimeError):
    """A persistent recurrence kernel gave up waiting for a workgroup that was not resident."""


class _ProjectedLstm(torch.autograd.Function):
    """x [B<=32, T, C] and the 5 parameters of every layer -> (y [B, T, proj], status).  Every tensor of the stack is
    [channels, T+1, BP] (slot 0 = the time before the first step)."""

    @staticmethod
    def forward(ctx, x, *params):
        B, T, C = x.shape
        L = len(params) // 5
        BP = (B + 15) // 16 * 16
        dev = x.device
        with torch.cuda.device(dev):
            seq = torch.zeros((C, T + 1, BP), dtype=torch.float32, device=dev)
            seq[:, 1:, :B] = x.detach().permute(2, 1, 0)
            status = torch.zeros((), dtype=torch.int32, device=dev)
            saved = []
            for layer in range(L):
                w_ih, w_hh, b_ih, b_hh, w_hr = (p.detach() for p in params[5 * layer:5 * layer + 5])
                H = w_hh.shape[0] // 4
                gates = torch.addmm((b_ih + b_hh)[:, None], w_ih, seq.view(seq.shape[0], -1)).view(4 * H, T + 1, BP)
                wrec = (w_hh.double() @ w_hr.double()).float().contiguous()
                hfull = torch.empty((H, T + 1, BP), dtype=torch.float32, device=dev)
                cell = torch.empty((H, T + 1, BP), dtype=torch.float32, device=dev)
                hfull[:, 0].zero_()
                cell[:, 0].zero_()
                status = status | _ops.lstm_recurrence_train(wrec, gates, hfull, cell)
                saved += [seq, gates, hfull, cell, wrec]
                seq = (w_hr @ hfull.view(H, -1)).view(w_hr.shape[0], T + 1, BP)
            out = seq[:, 1:, :B].permute(2, 1, 0).contiguous()
        ctx.save_for_backward(*params, *saved)
        ctx.dims = (B, T, BP, L)
        ctx.mark_non_differentiable(status)
        return out, status

    @staticmethod
    def backward(ctx, dout, _dstatus):
        B, T, BP, L = ctx.dims
        if getattr(ctx, "consumed", False):
            raise RuntimeError("unet_amd.lstm: this node's saved gates were turned into gradients by an earlier backward "
                               "pass; a second pass through the same graph (retain_graph) is not supported")
        ctx.consumed = True
        params, saved = ctx.saved_tensors[:5 * L], ctx.saved_tensors[5 * L:]
        dev = dout.device
        grads = [None] * (5 * L)
        with torch.cuda.device(dev):
            dy = torch.zeros((dout.shape[2], T + 1, BP), dtype=torch.float32, device=dev)
            dy[:, 1:, :B] = dout.permute(2, 1, 0)
            status = torch.zeros((), dtype=torch.int32, device=dev)
            for layer in reversed(range(L)):
                w_ih, w_hh, _, _, w_hr = (p.detach() for p in params[5 * layer:5 * layer + 5])
                seq, gates, hfull, cell, wrec = saved[5 * layer:5 * layer + 5]
                H = hfull.shape[0]
                Y, M, X = dy.view(dy.shape[0], -1), hfull.view(H, -1), seq.view(seq.shape[0], -1)
                dhfull = (w_hr.t() @ Y).view(H, T + 1, BP)
                d_w_hr = Y[:, BP:] @ M[:, BP:].t()                       # through y_t = W_hr m_t
                # the gates become the gradients of the gate pre-activations in place (a second backward through
                # this node would need them again: retain_graph is not supported)
                status = status | _ops.lstm_bptt(wrec, gates, cell, dhfull)
                G = gates.view(4 * H, -1)
                Gs = G[:, BP:]                                           # steps 0..T-1 (slot 0 is not a step)
                d_wrec = Gs @ M[:, :-BP].t()                             # G_t m_{t-1}^T
                grads[5 * layer + 0] = Gs @ X[:, BP:].t()
                grads[5 * layer + 1] = d_wrec @ w_hr.t()
                grads[5 * layer + 2] = Gs.sum(dim=1)
                grads[5 * layer + 3] = grads[5 * layer + 2].clone()
                grads[5 * layer + 4] = d_w_hr + w_hh.t() @ d_wrec
                dy = (w_ih.t() @ G).view(w_ih.shape[1], T + 1, BP)       # slot 0 holds no step and is never read
            dx = dy[:, 1:, :B].permute(2, 1, 0).contiguous()
            if int(status.item()) != 0:
                raise RecurrenceAborted("unet_amd.lstm: the reverse recurrence kernel gave up waiting for a workgroup "
                                        "that was not resident (GPU shared with other work?)")
        return (dx, *grads)


def lstm_forward_train(lstm, x):
    """x [B, T, input] -> [B, T, proj] with autograd history: gradients reach x and every parameter of ``lstm``.
    Zero initial state.  Falls back to torch's LSTM for a call whose forward kernel reports an abort."""
    if not trainable_on_device(lstm, x):
        raise NotImplementedError("lstm_forward_train: a projected fp32 LSTM of hidden size 640 or 300 on a 256-CU device")
    B = x.shape[0]
    if B > XCD_BATCH_PER_LAUNCH:                    # sequences are independent: run them in groups
        return torch.cat([lstm_forward_train(lstm, x[i:i + XCD_BATCH_PER_LAUNCH])
                          for i in range(0, B, XCD_BATCH_PER_LAUNCH)], dim=0)
    params = [getattr(lstm, name.format(layer)) for layer in range(lstm.num_layers) for name in PARAMS_PER_LAYER]
    out, status = _ProjectedLstm.apply(x.contiguous(), *params)
    if int(status.item()) != 0:
        import warnings
        warnings.warn("unet_amd.lstm: the XCD-local recurrence kernel gave up waiting for a workgroup that was not "
                      "resident (GPU shared with other work?); running this call on torch's LSTM")
        return lstm(x)[0]
    return out
