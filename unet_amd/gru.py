"""Single-layer unidirectional nn.GRU on the device (the TGRU bottlenecks, DSUNet.py:281-312 and UDSS.py:39-70, and
the four GRUs of UDSS.py:190-239's cRNN).

The input contribution of all time steps is one matrix product before the recurrence (a 1x1 convolution of the
library's own kernel); the recurrence itself is one
persistent HIP launch (unetmi_gru_recurrence) instead of several launches per time step.  Eval mode only (no
autograd graph is recorded); a GRU in train mode stays on torch's device RNN.
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


def supported(gru, x=None):
    ok = (isinstance(gru, nn.GRU) and gru.hidden_size in (128, 256, 512) and gru.num_layers == 1 and gru.batch_first
          and not gru.bidirectional and gru.bias and not gru.training)
    if ok and x is not None:
        ok = x.is_cuda and x.dim() == 3 and x.shape[0] <= MAX_BATCH_PER_LAUNCH * MAX_GROUPS
    return ok


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


def run(gru, x, h0=None):
    """nn.GRU's call: the persistent kernel where it applies, torch's device GRU otherwise."""
    if supported(gru, x) and x.dtype == torch.float32:
        return gru_forward(gru, x.contiguous(), h0)
    return gru(x, h0)
