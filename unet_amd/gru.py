"""Single-layer unidirectional nn.GRU on the device (the TGRU bottlenecks, DSUNet.py:281-312 and UDSS.py:39-70, and
the four GRUs of UDSS.py:190-239's cRNN).

The input contribution of all time steps is one matrix product before the recurrence (a 1x1 convolution of the
library's own kernel); the recurrence itself is one
persistent HIP launch (unetmi_gru_recurrence) instead of several launches per time step.  In eval mode no autograd
graph is recorded; a GRU in train mode is one autograd node (``gru_forward_train``): the forward recurrence keeps
the gates, backpropagation through time is one persistent launch as well (unetmi_gru_bptt), and the weight and input
gradients are time-batched matrix products over its result.
"""

import weakref

import torch
import torch.nn as nn

from . import ops as _ops

# The input product W_ih x of all steps: "conv" = a 1x1 convolution of the library's own kernel, "blas" = torch.matmul
GEMM = "conv"
_caches = weakref.WeakKeyDictionary()     # per GRU module: the packed input-weight image, rebuilt when a parameter changes

MAX_BATCH_PER_LAUNCH = 64
MAX_GROUPS = 8                    # beyond this many launches per call (batch > 512) the sequences are short and
                                  # numerous (the frequency-axis GRU): torch's batched RNN serves those


def _shape_supported(gru, x=None):
    ok = (isinstance(gru, nn.GRU) and gru.hidden_size in (128, 256, 512) and gru.num_layers == 1 and gru.batch_first
          and not gru.bidirectional and gru.bias)
    if ok and x is not None:
        ok = x.is_cuda and x.dim() == 3 and x.shape[0] <= MAX_BATCH_PER_LAUNCH * MAX_GROUPS
    return ok


def supported(gru, x=None):
    return _shape_supported(gru, x) and not gru.training


def gru_forward(gru, x, h0=None):
    """x [B, T, input] (+ h0 [1, B, H]) -> (y [B, T, H], h_n [1, B, H]) as nn.GRU returns them."""
    if not supported(gru):
        raise NotImplementedError("gru_forward: built for single-layer unidirectional batch_first GRUs with hidden "
                                  "size 128, 256 or 512, in eval mode")
    if not x.is_cuda or x.dtype != torch.float32:
        raise ValueError("gru_forward: x must be an fp32 device tensor")
    B, T, _ = x.shape
    if B > MAX_BATCH_PER_LAUNCH:                     # sequences are independent: run them in groups
        ys, hs = [], []
        for i in range(0, B, MAX_BATCH_PER_LAUNCH):
            y, h = gru_forward(gru, x[i:i + MAX_BATCH_PER_LAUNCH], None if h0 is None else h0[:, i:i + MAX_BATCH_PER_LAUNCH])
            ys.append(y)
            hs.append(h)
        return torch.cat(ys, dim=0), torch.cat(hs, dim=1)
    H = gru.hidden_size
    BP = (B + 15) // 16 * 16
    dev = x.device
    with torch.no_grad(), torch.cuda.device(dev):
        hfull = torch.empty((T + 1, H, BP), dtype=torch.float32, device=dev)
        hfull[0].zero_()
        if h0 is not None:
            hfull[0, :, :B] = h0[0].float().t()
        whh = gru.weight_hh_l0.detach().float().contiguous()
        bhh = gru.bias_hh_l0.detach().float().contiguous()
        if GEMM == "conv":
            # the input product as a 1x1 convolution of the library's own kernel over a [channels, T, BP] image; the
            # recurrence kernel reads its [3H, T, BP] result as it stands
            from .lstm import _pointwise
            seq = torch.zeros((1, x.shape[2], T, BP), dtype=torch.float32, device=dev)
            seq[0, :, :, :B] = x.permute(2, 1, 0)
            cache = _caches.setdefault(gru, {})
            gx = _pointwise(gru.weight_ih_l0, gru.bias_ih_l0, cache, "ih")(seq)[0]
            aborted = _ops.gru_recurrence(whh, bhh, gx, hfull, True)
        else:
            seq = torch.zeros((T, x.shape[2], BP), dtype=torch.float32, device=dev)       # [T, in, BP]
            seq[:, :, :B] = x.permute(1, 2, 0)
            gx = torch.matmul(gru.weight_ih_l0.detach().float(), seq) + gru.bias_ih_l0.detach().float()[None, :, None]
            aborted = _ops.gru_recurrence(whh, bhh, gx.contiguous(), hfull, False)
        y = hfull[1:, :, :B].permute(2, 0, 1).contiguous()
        # a recurrence that gave up waiting must not pass for a result: NaN, without a host round trip
        y = torch.where(aborted != 0, torch.full_like(y, float("nan")), y)
        return y, y[:, -1].unsqueeze(0).contiguous() if T > 0 else hfull[0, :, :B].t().unsqueeze(0).contiguous()


def wide_supported(gru, x=None):
    """A single-layer batch_first GRU (one or two directions) in eval mode, for wide_forward."""
    ok = isinstance(gru, nn.GRU) and gru.num_layers == 1 and gru.batch_first and gru.bias and not gru.training
    if ok and x is not None:
        ok = x.is_cuda and x.dim() == 3 and x.dtype == torch.float32
    return ok


def wide_forward(gru, x):
    """x [M, L, input] -> y [M, L, D*H] for MANY short sequences (the frequency-axis GRU of UDSS.py:15-37: M = B*T in
    the thousands, L = a few frequency rows, D = 2 directions): every step is one matrix product over all sequences
    (torch.matmul) and one launch for the gates (unetmi_gru_gate_fwd) -- the persistent recurrence kernel is built for
    the opposite case, a few long sequences.  Zero initial state, inference only; returns y alone."""
    if not wide_supported(gru, x):
        raise NotImplementedError("wide_forward: a single-layer batch_first fp32 GRU in eval mode on the device")
    M, L, _ = x.shape
    H, D = gru.hidden_size, 2 if gru.bidirectional else 1
    with torch.no_grad(), torch.cuda.device(x.device):
        names = [""] + (["_reverse"] if D == 2 else [])
        w_ih = torch.cat([getattr(gru, "weight_ih_l0" + n).detach().float() for n in names], dim=0)      # [D*3H, in]
        b_ih = torch.cat([getattr(gru, "bias_ih_l0" + n).detach().float() for n in names], dim=0)
        gx = torch.addmm(b_ih, x.reshape(M * L, -1), w_ih.t()).view(M, L, D * 3 * H)
        y = torch.empty((M, L, D * H), dtype=torch.float32, device=x.device)
        for d, n in enumerate(names):
            w_hh = getattr(gru, "weight_hh_l0" + n).detach().float()
            b_hh = getattr(gru, "bias_hh_l0" + n).detach().float().contiguous()
            order = range(L) if d == 0 else range(L - 1, -1, -1)
            prev = -1
            for step in order:
                gh = None if prev < 0 else torch.matmul(y[:, prev, d * H:(d + 1) * H], w_hh.t())
                _ops.gru_gate(gx, step, d, gh, b_hh, y, prev)
                prev = step
    return y


class RecurrenceAborted(RuntimeError):
    """A persistent recurrence kernel gave up waiting for a workgroup that was not resident."""


class _Gru(torch.autograd.Function):
    """(x [B<=64, T, C], h0 [B, H], w_ih, w_hh, b_ih, b_hh) -> (y [B, T, H], status); step-major tensors [T, rows, BP]."""

    @staticmethod
    def forward(ctx, x, h0, w_ih, w_hh, b_ih, b_hh):
        B, T, C = x.shape
        H = w_hh.shape[1]
        BP = (B + 15) // 16 * 16
        dev = x.device
        with torch.cuda.device(dev):
            seq = torch.zeros((T, C, BP), dtype=torch.float32, device=dev)
            seq[:, :, :B] = x.detach().permute(1, 2, 0)
            gx = (torch.matmul(w_ih.detach(), seq) + b_ih.detach()[None, :, None]).contiguous()
            hfull = torch.empty((T + 1, H, BP), dtype=torch.float32, device=dev)
            hfull[0].zero_()
            hfull[0, :, :B] = h0.detach().t()
            gates = torch.empty((T, 4 * H, BP), dtype=torch.float32, device=dev)
            status = _ops.gru_recurrence_train(w_hh.detach().contiguous(), b_hh.detach().contiguous(), gx, hfull, gates)
            y = hfull[1:, :, :B].permute(2, 0, 1).contiguous()
        ctx.save_for_backward(w_ih, w_hh, seq, hfull, gates)
        ctx.dims = (B, T, C, H, BP)
        ctx.mark_non_differentiable(status)
        return y, status

    @staticmethod
    def backward(ctx, dy, _dstatus):
        B, T, C, H, BP = ctx.dims
        if getattr(ctx, "consumed", False):
            raise RuntimeError("unet_amd.gru: this node's saved gates were turned into gradients by an earlier backward "
                               "pass; a second pass through the same graph (retain_graph) is not supported")
        ctx.consumed = True
        w_ih, w_hh, seq, hfull, gates = ctx.saved_tensors
        dev = dy.device
        with torch.cuda.device(dev):
            dyp = torch.zeros((T, H, BP), dtype=torch.float32, device=dev)
            dyp[:, :, :B] = dy.permute(1, 2, 0)
            # the gates become the pre-activation gradients in place (no second backward through this node)
            dh0, status = _ops.gru_bptt(w_hh.detach().contiguous(), gates, hfull, dyp)
            if int(status.item()) != 0:
                raise RecurrenceAborted("unet_amd.gru: the reverse recurrence kernel gave up waiting for a workgroup "
                                        "that was not resident (GPU shared with other work?)")
            G = gates.permute(1, 0, 2).reshape(4 * H, T * BP)
            hprev = hfull[:T].permute(1, 0, 2).reshape(H, T * BP)
            X = seq.permute(1, 0, 2).reshape(C, T * BP)
            g_hh = G[:3 * H]                                          # w.r.t. W_hh h + b_hh
            g_ih = torch.cat([G[:2 * H], G[3 * H:]], dim=0)           # w.r.t. W_ih x + b_ih
            d_w_hh, d_b_hh = g_hh @ hprev.t(), g_hh.sum(dim=1)
            d_w_ih, d_b_ih = g_ih @ X.t(), g_ih.sum(dim=1)
            dx = (w_ih.detach().t() @ g_ih).view(C, T, BP)[:, :, :B].permute(2, 1, 0).contiguous()
        return dx, dh0[:, :B].t().contiguous(), d_w_ih, d_w_hh, d_b_ih, d_b_hh


def gru_forward_train(gru, x, h0=None):
    """nn.GRU's call with autograd history on the persistent kernels: gradients reach x, h0 and the four parameters."""
    if not (_shape_supported(gru, x) and x.dtype == torch.float32):
        raise NotImplementedError("gru_forward_train: a single-layer unidirectional batch_first fp32 GRU with hidden "
                                  "size 128, 256 or 512 on the device")
    B = x.shape[0]
    H = gru.hidden_size
    if B > MAX_BATCH_PER_LAUNCH:
        ys, hs = [], []
        for i in range(0, B, MAX_BATCH_PER_LAUNCH):
            y, h = gru_forward_train(gru, x[i:i + MAX_BATCH_PER_LAUNCH],
                                     None if h0 is None else h0[:, i:i + MAX_BATCH_PER_LAUNCH])
            ys.append(y)
            hs.append(h)
        return torch.cat(ys, dim=0), torch.cat(hs, dim=1)
    h_init = torch.zeros((B, H), dtype=torch.float32, device=x.device) if h0 is None else h0[0].float()
    y, status = _Gru.apply(x.contiguous(), h_init, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    if int(status.item()) != 0:
        import warnings
        warnings.warn("unet_amd.gru: the recurrence kernel gave up waiting for a workgroup that was not resident (GPU "
                      "shared with other work?); running this call on torch's GRU")
        return gru(x, h0)
    h_n = y[:, -1].unsqueeze(0) if x.shape[1] > 0 else h_init.unsqueeze(0)
    return y, h_n


def run(gru, x, h0=None):
    """nn.GRU's call: the persistent kernels where they apply (eval: gru_forward; train: the autograd node
    gru_forward_train), torch's device GRU otherwise."""
    if x.dtype == torch.float32:
        if supported(gru, x):
            return gru_forward(gru, x.contiguous(), h0)
        if gru.training and torch.is_grad_enabled() and _shape_supported(gru, x):
            return gru_forward_train(gru, x, h0)
    return gru(x, h0)
