"""STFT / inverse STFT on the device with the convolution kernels.

The reference's helpers call ``torch.stft(x, n_fft, window=hann)`` / ``torch.istft`` with their
defaults (DSUNet.py:693,728: hop n_fft/4, centred, reflect padding, one-sided).  With hop = n_fft/4
a frame is four consecutive hop-sized blocks, so writing the padded signal as a [hop]-channel
sequence of blocks turns the windowed DFT of every frame into a convolution with kernel width 4:

    analysis   Conv2d(hop -> 2F, (1,4))            weights = window * cos / -sin
    synthesis  ConvTranspose2d(2F -> hop, (1,4))   weights = window * inverse-DFT basis (overlap-add),
               followed by the division by the summed squared window

Both run on the MFMA convolution kernel (exact fp32 products).  The signals of a batch are laid
out along the H axis of one image, so the kernel's row tiles are full.
"""

import math

import torch

from . import _lib
from .ops import conv2d_fwd, pack_conv_weights


class DeviceSTFT:
    def __init__(self, n_fft=512, window=None):
        if n_fft % 4 != 0:
            raise ValueError("n_fft must be a multiple of 4")
        self.n_fft, self.hop, self.n_freq = n_fft, n_fft // 4, n_fft // 2 + 1
        self.window = (torch.hann_window(n_fft) if window is None else window.detach().cpu()).double()
        self._dev = {}

    # -- weights ---------------------------------------------------------------------------
    def _build(self, device):
        N, hop, F = self.n_fft, self.hop, self.n_freq
        n = torch.arange(N, dtype=torch.float64)
        f = torch.arange(F, dtype=torch.float64)
        ang = 2.0 * math.pi * f[:, None] * n[None, :] / N                       # [F, N]
        w = self.window
        # analysis: rows 0..F-1 real part, F..2F-1 imaginary part; sample n = q*hop + j -> weight [k, j, 0, q]
        ana = torch.cat([torch.cos(ang) * w, -torch.sin(ang) * w], dim=0)       # [2F, N]
        ana = ana.reshape(2 * F, 4, hop).permute(0, 2, 1).reshape(2 * F, hop, 1, 4).float().contiguous()
        # synthesis (torch ConvTranspose2d layout [C_in = 2F, C_out = hop, 1, 4]):
        # frame[n] = (1/N) sum_k c_k (Re Y_k cos - Im Y_k sin), c_k = 1 at DC and Nyquist else 2, times the window
        ck = torch.full((F,), 2.0, dtype=torch.float64)
        ck[0] = 1.0
        ck[-1] = 1.0
        syn_re = ck[:, None] * torch.cos(ang) * w / N
        syn_im = -ck[:, None] * torch.sin(ang) * w / N
        syn_im[0] = 0.0                                                         # irfft ignores these two
        syn_im[-1] = 0.0
        syn = torch.cat([syn_re, syn_im], dim=0)                                # [2F, N]
        syn = syn.reshape(2 * F, 4, hop).permute(0, 2, 1).reshape(2 * F, hop, 1, 4).float().contiguous()
        geom_a = [2 * F, 1, 4, 1, 1, 0, 0, 1, 1, 0, 0]
        geom_s = [hop, 1, 4, 1, 1, 0, 0, 1, 1, 0, 0]
        f32 = dict(dtype=torch.float32, device=device)
        st = dict(
            geom_a=geom_a, geom_s=geom_s,
            w_ana=pack_conv_weights(ana, hop, 0, geom_a, [0, _lib.IN_SINGLE, _lib.ACT_NONE, 0]).to(device),
            w_syn=pack_conv_weights(syn, 2 * F, 0, geom_s, [1, _lib.IN_SINGLE, _lib.ACT_NONE, 0]).to(device),
            zero_a=torch.zeros(2 * F, **f32), one_a=torch.ones(2 * F, **f32),
            zero_s=torch.zeros(hop, **f32), one_s=torch.ones(hop, **f32),
            wsq=(w * w).float().to(device))
        self._dev[str(device)] = st
        return st

    def _state(self, device):
        return self._dev.get(str(device)) or self._build(device)

    # -- transforms ------------------------------------------------------------------------
    def stft(self, x):
        """x [S, L] fp32 device -> complex64 [S, F, T], T = 1 + L // hop (torch.stft defaults)."""
        if x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("stft: x must be a 2-D fp32 device tensor [signals, samples]")
        st = self._state(x.device)
        S, L = x.shape
        hop, F = self.hop, self.n_freq
        T = 1 + L // hop
        xp = torch.nn.functional.pad(x[:, None, :], (self.n_fft // 2, self.n_fft // 2), mode="reflect")[:, 0]
        need = (T + 3) * hop
        xp = xp[:, :need] if xp.shape[1] >= need else torch.nn.functional.pad(xp, (0, need - xp.shape[1]))
        blocks = xp.reshape(S, T + 3, hop).permute(2, 0, 1).contiguous()[None]   # [1, hop, S, T+3]
        out = conv2d_fwd(blocks, None, st["w_ana"], st["zero_a"], st["one_a"], st["zero_a"], None, None, None,
                         st["geom_a"], [0, _lib.IN_SINGLE, _lib.ACT_NONE, 0])     # [1, 2F, S, T]
        ri = out[0].reshape(2, F, S, T).permute(2, 1, 3, 0).contiguous()         # [S, F, T, 2]
        return torch.view_as_complex(ri)

    def istft(self, Y, length):
        """Y complex64 [S, F, T] -> [S, length] (torch.istft defaults: centred, window envelope removed)."""
        if Y.dim() != 3 or not Y.is_cuda or Y.dtype != torch.complex64:
            raise ValueError("istft: Y must be a 3-D complex64 device tensor [signals, F, T]")
        st = self._state(Y.device)
        S, F, T = Y.shape
        hop = self.hop
        ri = torch.view_as_real(Y).permute(3, 1, 0, 2).reshape(1, 2 * F, S, T).contiguous()
        out = conv2d_fwd(ri, None, st["w_syn"], st["zero_s"], st["one_s"], st["zero_s"], None, None, None,
                         st["geom_s"], [1, _lib.IN_SINGLE, _lib.ACT_NONE, 0])     # [1, hop, S, T+3]
        y = out[0].permute(1, 2, 0).reshape(S, (T + 3) * hop)
        # summed squared window over the frames that cover each sample
        env = torch.zeros((T + 3) * hop, dtype=torch.float32, device=Y.device)
        wsq = st["wsq"].reshape(4, hop)
        for q in range(4):
            env.view(T + 3, hop)[q:q + T] += wsq[q]
        start = self.n_fft // 2
        end = min(start + length, y.shape[1])
        y = y[:, start:end] / env[start:end]
        if y.shape[1] < length:
            y = torch.nn.functional.pad(y, (0, length - y.shape[1]))
        return y.contiguous()
