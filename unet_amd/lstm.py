"""Projected LSTM on the device (the bottleneck of ResUNet, ResUNet.py:139,186-197).

Everything without a time dependency is a large GEMM (torch.matmul = the ROCm BLAS library):
the input contribution of all time steps before the recurrence, the projection of all
pre-projection states after it.  The recurrence itself is one persistent HIP launch per
layer (unetmi_lstm_recurrence) instead of three launches per time step.
"""

import torch
import torch.nn as nn

from . import _lib
from ._lib import check

MAX_BATCH_PER_LAUNCH = 64


def supported(lstm):
    return (isinstance(lstm, nn.LSTM) and lstm.hidden_size == 640 and lstm.proj_size > 0 and lstm.batch_first
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


def lstm_forward(lstm, x, cache):
    """x [B, T, input] -> [B, T, proj]; zero initial state, inference only."""
    if not supported(lstm):
        raise NotImplementedError("lstm_forward: built for batch_first projected LSTMs with hidden size 640")
    if not x.is_cuda or x.dtype != torch.float32:
        raise ValueError("lstm_forward: x must be an fp32 device tensor")
    B, T, _ = x.shape
    if B > MAX_BATCH_PER_LAUNCH:                    # sequences are independent: run them in groups
        return torch.cat([lstm_forward(lstm, x[i:i + MAX_BATCH_PER_LAUNCH], cache)
                          for i in range(0, B, MAX_BATCH_PER_LAUNCH)], dim=0)
    H = lstm.hidden_size
    BP = (B + 15) // 16 * 16
    dev = x.device
    with torch.no_grad():
        seq = torch.zeros((T, x.shape[2], BP), dtype=torch.float32, device=dev)      # [T, in, BP]
        seq[:, :, :B] = x.permute(1, 2, 0)
        sync = torch.empty(lstm.hidden_size // 4 + 1, dtype=torch.int32, device=dev)
        aborted = torch.zeros((), dtype=torch.int32, device=dev)
        for layer in range(lstm.num_layers):
            w_ih = getattr(lstm, "weight_ih_l{}".format(layer)).detach().float()
            w_hr = getattr(lstm, "weight_hr_l{}".format(layer)).detach().float()
            bias = (getattr(lstm, "bias_ih_l{}".format(layer)).float() + getattr(lstm, "bias_hh_l{}".format(layer)).float()).detach()
            gx = torch.matmul(w_ih, seq) + bias[None, :, None]                        # [T, 4H, BP]
            wrec = _folded_recurrent_matrix(lstm, layer, cache)
            hfull = torch.empty((T + 1, H, BP), dtype=torch.float32, device=dev)
            hfull[0].zero_()
            with torch.cuda.device(dev):
                check(_lib.lib.unetmi_lstm_recurrence(T, BP, H, wrec.data_ptr(), gx.data_ptr(), hfull.data_ptr(),
                                                      sync.data_ptr(), torch.cuda.current_stream().cuda_stream),
                      "unetmi_lstm_recurrence")
            aborted = aborted | sync[-1]                                              # the kernel's status word, kept on the device
            seq = torch.matmul(w_hr, hfull[1:])                                       # [T, proj, BP] = next layer's input
        out = seq[:, :, :B].permute(2, 0, 1).contiguous()
        # a recurrence whose device-wide barrier timed out (GPU shared with other work) must not pass for a
        # result: the whole output becomes NaN, without a host round trip
        return torch.where(aborted != 0, torch.full_like(out, float("nan")), out)
