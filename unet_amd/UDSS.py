"""Device ``UDSS`` -- the complex-valued directional source separation U-Net -- with the reference's
constructor, ``forward(sf, SV, theta)`` and state_dict keys (UDSS.py:420-683; complex blocks UNet_m.py:88-132,
:214-225).

A complex convolution is four real convolutions that share their input (UNet_m.py:98-102).  On the device a
``[B, C, F, T, 2]`` tensor is kept as the real tensor ``[B, 2C, F, T]`` (real parts of all channels, then the
imaginary parts), and a complex layer is ONE launch of the real convolution kernel with the weight

        | W_re  -W_im |            bias  | b_re - b_im |
        | W_im   W_re |                  | b_re + b_im |

its two BatchNorms side by side in the epilogue, the activation, the attractor gate (the same factor for both
parts) and -- for decoders -- the concat with the skip tensor read as a second input.  The stacked weight is
rebuilt when a parameter changes.  Only the network input and nothing else is re-laid out.
"""

import torch
import torch.nn as nn

from .DSUNet import Attractor, AttractEncoder
from . import gru as _gru
from .fused import FusedConv, require_device_tensor


def stack_complex(x):
    """[B, C, F, T, 2] -> [B, 2C, F, T] (real parts first)."""
    B, C, F, T, _ = x.shape
    return x.permute(0, 4, 1, 2, 3).reshape(B, 2 * C, F, T).contiguous()


def unstack_complex(x):
    """[B, 2C, F, T] -> [B, C, F, T, 2]."""
    B, C2, F, T = x.shape
    return x.reshape(B, 2, C2 // 2, F, T).permute(0, 2, 3, 4, 1).contiguous()


class ComplexConv2d(nn.Module):
    """UNet_m.py:88-102.  ``forward`` takes and returns ``[B, C, F, T, 2]``."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 **kwargs):
        super().__init__()
        kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias, **kwargs)
        self.conv_re = nn.Conv2d(in_channel, out_channel, kernel_size, **kw)
        self.conv_im = nn.Conv2d(in_channel, out_channel, kernel_size, **kw)
        self._layer = None

    def parts(self):
        return self.conv_re, self.conv_im

    def forward(self, x):
        require_device_tensor(x, "ComplexConv2d")
        if self._layer is None:
            object.__setattr__(self, "_layer", _ComplexLayer(self, None, None))
        return unstack_complex(self._layer(stack_complex(x)))


class ComplexConvTranspose2d(nn.Module):
    """UNet_m.py:104-132."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, bias=True, **kwargs):
        super().__init__()
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding, output_padding=output_padding,
                  groups=groups, bias=bias, dilation=dilation, **kwargs)
        self.tconv_re = nn.ConvTranspose2d(in_channel, out_channel, **kw)
        self.tconv_im = nn.ConvTranspose2d(in_channel, out_channel, **kw)
        self._layer = None

    def parts(self):
        return self.tconv_re, self.tconv_im

    def forward(self, x):
        require_device_tensor(x, "ComplexConvTranspose2d")
        if self._layer is None:
            object.__setattr__(self, "_layer", _ComplexLayer(self, None, None))
        return unstack_complex(self._layer(stack_complex(x)))


class ComplexBatchNorm2d(nn.Module):
    """UNet_m.py:214-225: independent BatchNorm2d on the two parts (torch ops when called on its own)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, **kwargs):
        super().__init__()
        kw = dict(num_features=num_features, momentum=momentum, affine=affine, eps=eps,
                  track_running_stats=track_running_stats, **kwargs)
        self.bn_re = nn.BatchNorm2d(**kw)
        self.bn_im = nn.BatchNorm2d(**kw)

    def forward(self, x):
        return torch.stack((self.bn_re(x[..., 0]), self.bn_im(x[..., 1])), dim=-1)


class ComplexInstanceNorm2d(nn.Module):
    """UNet_m.py:227-237."""

    def __init__(self, num_features):
        super().__init__()
        self.in_re = nn.InstanceNorm2d(num_features=num_features)
        self.in_im = nn.InstanceNorm2d(num_features=num_features)

    def forward(self, x):
        return torch.stack((self.in_re(x[..., 0]), self.in_im(x[..., 1])), dim=-1)


class _ComplexLayer:
    """One complex conv / transposed conv [+ ComplexBatchNorm2d] [+ activation] on stacked tensors as one
    fused real-convolution launch.  Owns only caches: the stacked twin modules, rebuilt per input split
    (channels of x, channels of the skip) and whenever a source parameter changes."""

    def __init__(self, cconv, cbn, act):
        self.cconv, self.cbn, self.act = cconv, cbn, act
        self._twins = {}
        self._key = None

    def _versions(self):
        ts = []
        for m in self.cconv.parts():
            ts += [m.weight, m.bias]
        if self.cbn is not None:
            for m in (self.cbn.bn_re, self.cbn.bn_im):
                ts += [m.weight, m.bias, m.running_mean, m.running_var]
        return tuple((t._version, t.data_ptr()) if t is not None else None for t in ts)

    def _twin(self, C0, C1):
        key = self._versions()
        if key != self._key:
            self._twins.clear()
            self._key = key
        fused = self._twins.get((C0, C1))
        if fused is not None:
            return fused
        re, im = self.cconv.parts()
        transposed = isinstance(re, nn.ConvTranspose2d)
        if re.groups != 1:
            raise NotImplementedError("grouped complex convolutions are not built")
        ia, oa = (0, 1) if transposed else (1, 0)                       # input / output channel axes of the weight
        Wr, Wi = re.weight.detach(), im.weight.detach()
        if Wr.shape[ia] != C0 + C1:
            raise ValueError("complex layer built for {} input channels, got {} + {}".format(Wr.shape[ia], C0, C1))
        Wr0, Wr1 = Wr.narrow(ia, 0, C0), Wr.narrow(ia, C0, C1)
        Wi0, Wi1 = Wi.narrow(ia, 0, C0), Wi.narrow(ia, C0, C1)
        # stacked input order: re(x), im(x), re(skip), im(skip); stacked output order: re, im
        to_re = torch.cat([Wr0, -Wi0, Wr1, -Wi1], dim=ia)
        to_im = torch.cat([Wi0, Wr0, Wi1, Wr1], dim=ia)
        W = torch.cat([to_re, to_im], dim=oa)
        K = Wr.shape[oa]
        kw = dict(kernel_size=re.kernel_size, stride=re.stride, padding=re.padding, dilation=re.dilation)
        if transposed:
            twin = nn.ConvTranspose2d(2 * (C0 + C1), 2 * K, output_padding=re.output_padding, **kw)
        else:
            twin = nn.Conv2d(2 * (C0 + C1), 2 * K, padding_mode=re.padding_mode, **kw)
        norm = None
        with torch.no_grad():
            twin = twin.to(W.device)
            twin.weight.copy_(W)
            br = re.bias.detach() if re.bias is not None else torch.zeros(K, device=W.device)
            bi = im.bias.detach() if im.bias is not None else torch.zeros(K, device=W.device)
            twin.bias.copy_(torch.cat([br - bi, br + bi]))
            if self.cbn is not None:
                a, b = self.cbn.bn_re, self.cbn.bn_im
                norm = nn.BatchNorm2d(2 * K, eps=a.eps).to(W.device).eval()
                norm.weight.copy_(torch.cat([a.weight, b.weight]))
                norm.bias.copy_(torch.cat([a.bias, b.bias]))
                norm.running_mean.copy_(torch.cat([a.running_mean, b.running_mean]))
                norm.running_var.copy_(torch.cat([a.running_var, b.running_var]))
        fused = FusedConv(twin, norm, self.act)
        if not fused.fusable():
            raise NotImplementedError("complex layer: this activation / norm is not one the epilogue has")
        fused.twin_modules = (twin, norm)             # the bfloat16-operand path builds its images from the same twin
        self._twins[(C0, C1)] = fused
        return fused

    def blocked(self, x, skip=None, gate=None, nchw_out=False, halves=1, out_fmt="bf16"):
        """The same layer on channel-blocked 16-bit tensors (unet_amd/blocked.py; operands in the input's format,
        output stored as ``out_fmt``).  The 16-bit kernel takes at
        most 128 output channels per launch; a wider layer (the 128-complex-channel bottleneck encoder) runs as
        two launches, real outputs and imaginary outputs, whose blocks are then joined."""
        from .blocked import BlockedConv, BlockedTensor
        C0 = x.C // 2
        C1 = 0 if skip is None else skip.C // 2
        fused = self._twin(C0, C1)
        layers = getattr(fused, "blocked_layers", None)
        if layers is None or fused.blocked_halves != (halves, out_fmt):
            twin, norm = fused.twin_modules
            K2 = twin.out_channels
            if K2 <= 128:
                layers = [BlockedConv(twin, norm, self.act, halves=halves, nchw_out=nchw_out, out_fmt=out_fmt)]
            else:
                transposed = isinstance(twin, nn.ConvTranspose2d)
                K = K2 // 2
                if K > 128 or K % 16 != 0:
                    raise NotImplementedError("complex layer with {} outputs on the bfloat16 path".format(K))
                layers = []
                for h in range(2):                         # real outputs, imaginary outputs
                    kw = dict(kernel_size=twin.kernel_size, stride=twin.stride, padding=twin.padding, dilation=twin.dilation)
                    part = (nn.ConvTranspose2d(twin.in_channels, K, output_padding=twin.output_padding, **kw) if transposed
                            else nn.Conv2d(twin.in_channels, K, **kw)).to(twin.weight.device)
                    pn = None
                    with torch.no_grad():
                        part.weight.copy_(twin.weight.narrow(1 if transposed else 0, h * K, K))
                        part.bias.copy_(twin.bias[h * K:(h + 1) * K])
                        if norm is not None:
                            pn = nn.BatchNorm2d(K, eps=norm.eps).to(twin.weight.device).eval()
                            for name in ("weight", "bias", "running_mean", "running_var"):
                                getattr(pn, name).copy_(getattr(norm, name)[h * K:(h + 1) * K])
                    layers.append(BlockedConv(part, pn, self.act, halves=halves, nchw_out=nchw_out, out_fmt=out_fmt))
            fused.blocked_layers = layers
            fused.blocked_halves = (halves, out_fmt)
        if len(layers) == 1:
            g = None if gate is None else torch.cat([gate, gate], dim=1).contiguous()
            return layers[0](x, skip, g)
        outs = [l(x, skip, None if gate is None else gate.contiguous()) for l in layers]
        return BlockedTensor(torch.cat([o.data for o in outs], dim=1), sum(o.C for o in outs))

    def __call__(self, x, skip=None, gate=None):
        """x, skip: stacked [B, 2C, F, T]; gate: [B, K] (applied to both parts)."""
        C0 = x.shape[1] // 2
        C1 = 0 if skip is None else skip.shape[1] // 2
        if gate is not None:
            gate = torch.cat([gate, gate], dim=1).contiguous()
        return self._twin(C0, C1)(x, skip, gate=gate)


def _complex_train(cconv, cbn, act, x, skip=None, gate=None, dropout=None):
    """A complex block on stacked tensors in train() mode, under autograd: the block weight
    [[W_re, -W_im], [W_im, W_re]] and the bias pair are assembled from the parameters by torch operations (so the
    gradient of the stacked weight flows back to the two real convolutions' parameters), and the convolution, the
    two BatchNorm2d (batch statistics; stacked they are one BatchNorm over 2K channels) and an activation known by
    code are one ``unetmi::conv_norm_act`` node -- forward, input gradient and weight gradient on the HIP kernels.
    A learnable activation (PReLU), dropout between convolution and norm, and the gate are torch operations."""
    from . import _lib
    from . import ops as _ops
    from .fused import activation_code
    from .train import _bn_bookkeeping  # noqa: F401  (same bookkeeping, applied per part below)
    re, im = cconv.parts()
    transposed = isinstance(re, nn.ConvTranspose2d)
    if re.groups != 1:
        raise NotImplementedError("grouped complex convolutions are not built")
    if cbn is not None and not isinstance(cbn, ComplexBatchNorm2d):
        raise NotImplementedError("complex block: train mode is built for ComplexBatchNorm2d")
    ia, oa = (0, 1) if transposed else (1, 0)
    C0 = x.shape[1] // 2
    C1 = 0 if skip is None else skip.shape[1] // 2
    Wr, Wi = re.weight, im.weight
    if Wr.shape[ia] != C0 + C1:
        raise ValueError("complex layer built for {} input channels, got {} + {}".format(Wr.shape[ia], C0, C1))
    Wr0, Wr1 = Wr.narrow(ia, 0, C0), Wr.narrow(ia, C0, C1)
    Wi0, Wi1 = Wi.narrow(ia, 0, C0), Wi.narrow(ia, C0, C1)
    W = torch.cat([torch.cat([Wr0, -Wi0, Wr1, -Wi1], dim=ia), torch.cat([Wi0, Wr0, Wi1, Wr1], dim=ia)], dim=oa).contiguous()
    K = Wr.shape[oa]
    zero = torch.zeros(K, device=W.device)
    br = re.bias if re.bias is not None else zero
    bi = im.bias if im.bias is not None else zero
    bias = torch.cat([br - bi, br + bi])
    geom, _ = FusedConv(re).geometry()
    geom = [2 * K] + list(geom[1:])
    found = activation_code(act)
    by_code = found is not None and not isinstance(found[1], torch.Tensor)
    code, par = found if by_code else (_lib.ACT_NONE, None)
    act_param = None if par is None else torch.full((1,), par, dtype=torch.float32, device=x.device)
    dropping = dropout is not None and dropout.p != 0.0
    parts = () if cbn is None else (cbn.bn_re, cbn.bn_im)
    for bn in parts:
        if not bn.track_running_stats or bn.momentum is None or not bn.affine:
            raise NotImplementedError("complex block: train mode needs affine BatchNorm2d with tracked statistics")
    gamma = torch.cat([bn.weight for bn in parts]) if parts else None
    beta = torch.cat([bn.bias for bn in parts]) if parts else None
    eps = parts[0].eps if parts else 0.0
    x = x.contiguous()
    skip = None if skip is None else skip.contiguous()
    if parts and not dropping:
        y, z, mean, invstd = torch.ops.unetmi.conv_norm_act(x, skip, W, bias, gamma, beta, None, None, geom, transposed,
                                                            1, eps, code, act_param)
    else:
        z = torch.ops.unetmi.conv_norm_act(x, skip, W, bias, None, None, None, None, geom, transposed, 0, 0.0,
                                           code if not parts else _lib.ACT_NONE, act_param if not parts else None)[0]
        y = z
        if dropping:
            z = dropout(z)
        if parts:
            y, mean, invstd = torch.ops.unetmi.bn_act(z, gamma, beta, None, None, 1, eps, code, act_param)
    if parts:
        with torch.no_grad():
            count = z.numel() // z.shape[1]
            for h, bn in enumerate(parts):
                _ops.bn_running_update(bn.running_mean, bn.running_var, mean[h * K:(h + 1) * K].contiguous(),
                                       invstd[h * K:(h + 1) * K].contiguous(), count, bn.eps, bn.momentum)
                torch.autograd.graph.increment_version(bn.running_mean)
                torch.autograd.graph.increment_version(bn.running_var)
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
    if not by_code:
        y = act(y)
    if gate is not None:
        y = y * torch.cat([gate, gate], dim=1)[:, :, None, None]
    return y


def _norm_cls(type_norm):
    # UDSS.py:358-361: any other name leaves the class unbound in the reference (UnboundLocalError)
    if type_norm == "BatchNorm2d":
        return ComplexBatchNorm2d
    if type_norm == "InstanceNorm2d":
        return ComplexInstanceNorm2d
    raise UnboundLocalError("local variable 'bn' referenced before assignment")


def _activation(name):
    return {"LeakyReLU": lambda: nn.LeakyReLU(inplace=True), "PReLU": nn.PReLU, "SiLU": nn.SiLU}[name]()


class _ComplexBlock(nn.Module):
    def _stacked(self, cconv, x, skip=None, gate=None):
        if self.training:
            return _complex_train(cconv, self.bn, self.relu, x, skip, gate, getattr(self, "dr", None))
        if not isinstance(self.bn, ComplexBatchNorm2d):
            raise NotImplementedError("{}: only ComplexBatchNorm2d (eval) is fused".format(type(self).__name__))
        if self._layer is None:
            object.__setattr__(self, "_layer", _ComplexLayer(cconv, self.bn, self.relu))
        return self._layer(x, skip, gate)


class Encoder(_ComplexBlock):
    """Complex conv -> dropout -> complex norm -> activation (UDSS.py:349-386, complex=True)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None, complex=True,
                 padding_mode="zeros", activation="LeakyReLU", dropout=0.0, type_norm="BatchNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is unet_amd.DSUNet")
        if padding is None:
            padding = [(k - 1) // 2 for k in kernel_size]
        self.dr = nn.Dropout(dropout)
        self.conv = ComplexConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                  padding=padding, padding_mode=padding_mode)
        self.bn = _norm_cls(type_norm)(out_channels)
        self.relu = _activation(activation)
        self._layer = None

    def forward_stacked(self, x, gate=None):
        return self._stacked(self.conv, x, None, gate)

    def forward(self, x):
        require_device_tensor(x, "UDSS.Encoder")
        return unstack_complex(self.forward_stacked(stack_complex(x)))


class Decoder(_ComplexBlock):
    """Complex transposed conv -> complex norm -> activation (UDSS.py:388-418, complex=True)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, output_padding, padding=(0, 0), complex=True,
                 activation="LeakyReLU", type_norm="BatchNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is unet_amd.DSUNet")
        self.transconv = ComplexConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                                output_padding=output_padding, padding=padding)
        self.bn = _norm_cls(type_norm)(out_channels)
        self.relu = _activation(activation)
        self._layer = None

    def forward_stacked(self, x, skip=None):
        return self._stacked(self.transconv, x, skip)

    def forward(self, x):
        require_device_tensor(x, "UDSS.Decoder")
        return unstack_complex(self.forward_stacked(stack_complex(x)))


class TGRUBlock(_ComplexBlock):
    """GRU along time over the interleaved (channel, re/im) features of every (batch, frequency) row (one persistent
    recurrence launch, unet_amd.gru), then complex 1x1 conv + complex BatchNorm + ReLU as one launch (UDSS.py:39-70)."""

    def __init__(self, in_channels, hidden_size, out_channels, skipGRU=False, **kwargs):
        super().__init__()
        if skipGRU:
            raise Exception("Not Implemented")
        self.GRU = nn.GRU(in_channels * 2, hidden_size * 2, batch_first=True)
        self.conv = ComplexConv2d(hidden_size, out_channels, kernel_size=1)
        self.bn = ComplexBatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.ReLU()
        self._layer = None

    def forward_stacked(self, x, rnn_state=None):
        B, C2, F, T = x.shape
        C = C2 // 2
        seq = x.reshape(B, 2, C, F, T).permute(0, 3, 4, 2, 1).reshape(B * F, T, C2)      # feature index c*2 + re/im
        y, _ = _gru.run(self.GRU, seq.float(), rnn_state)
        H = self.hidden_size
        y = y.reshape(B, F, T, H, 2).permute(0, 4, 3, 1, 2).reshape(B, 2 * H, F, T).contiguous().to(x.dtype)
        return self._stacked(self.conv, y)

    def forward(self, x, rnn_state=None):
        require_device_tensor(x, "UDSS.TGRUBlock")
        return unstack_complex(self.forward_stacked(stack_complex(x), rnn_state))


class FGRUBlock(_ComplexBlock):
    """Bidirectional GRU along frequency for every (batch, frame) (unet_amd.gru.wide_forward in eval mode: a matrix
    product and one gate launch per step and direction; torch's device GRU under autograd), then complex 1x1 conv +
    complex BatchNorm + PReLU as one launch (UDSS.py:15-37)."""

    def __init__(self, in_channels, hidden_size, out_channels):
        super().__init__()
        self.GRU = nn.GRU(in_channels * 2, hidden_size * 2, batch_first=True, bidirectional=True)
        self.conv = ComplexConv2d(hidden_size * 2, out_channels, kernel_size=1)
        self.bn = ComplexBatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.PReLU()
        self._layer = None

    def forward_stacked(self, x):
        B, C2, F, T = x.shape
        C = C2 // 2
        seq = x.reshape(B, 2, C, F, T).permute(0, 4, 3, 2, 1).reshape(B * T, F, C2)      # feature index c*2 + re/im
        seq = seq.float().contiguous()
        if _gru.wide_supported(self.GRU, seq):          # eval: one product + one gate launch per step and direction
            y = _gru.wide_forward(self.GRU, seq)
        else:
            y, _ = self.GRU(seq)
        H2 = self.hidden_size * 2
        y = y.reshape(B, T, F, H2, 2).permute(0, 4, 3, 2, 1).reshape(B, 2 * H2, F, T).contiguous().to(x.dtype)
        return self._stacked(self.conv, y)

    def forward(self, x):
        require_device_tensor(x, "UDSS.FGRUBlock")
        return unstack_complex(self.forward_stacked(stack_complex(x)))


class FTGRUBlock(nn.Module):
    """UDSS.py:72-83."""

    def __init__(self, in_channels, hidden_size, out_channels, **kwargs):
        super().__init__()
        self.FGRU = FGRUBlock(in_channels, hidden_size, out_channels, **kwargs)
        self.TGRU = TGRUBlock(in_channels, hidden_size, out_channels, **kwargs)

    def forward_stacked(self, x, rnn_state=None):
        return self.TGRU.forward_stacked(self.FGRU.forward_stacked(x), rnn_state)

    def forward(self, x, rnn_state=None):
        require_device_tensor(x, "UDSS.FTGRUBlock")
        return unstack_complex(self.forward_stacked(stack_complex(x), rnn_state))


class cRNN(nn.Module):
    """Four real GRUs combined as a complex recurrence, BatchNorm1d and PReLU per part (UDSS.py:190-239); the GRUs
    through unet_amd.gru, elementwise ops on the stacked tensor."""

    def __init__(self, dim, style="GRU"):
        super().__init__()
        cell = {"GRU": nn.GRU, "LSTM": nn.LSTM}[style]
        self.rr = cell(dim, dim, batch_first=True)
        self.ri = cell(dim, dim, batch_first=True)
        self.ir = cell(dim, dim, batch_first=True)
        self.ii = cell(dim, dim, batch_first=True)
        self.re_norm = nn.BatchNorm1d(dim)
        self.im_norm = nn.BatchNorm1d(dim)
        self.re_activation = nn.PReLU()
        self.im_activation = nn.PReLU()

    def forward_stacked(self, x):
        B, C2, F, T = x.shape
        C = C2 // 2
        parts = x.float().reshape(B, 2, C * F, T).permute(1, 0, 3, 2)                      # [2, B, T, C*F]
        cell = _gru.run if isinstance(self.rr, nn.GRU) else (lambda rnn, seq: rnn(seq))
        rr, ri = cell(self.rr, parts[0])[0], cell(self.ri, parts[0])[0]
        ir, ii = cell(self.ir, parts[1])[0], cell(self.ii, parts[1])[0]
        re = self.re_activation(self.re_norm(torch.permute(rr - ii, (0, 2, 1))))           # [B, C*F, T]
        im = self.im_activation(self.im_norm(torch.permute(ri + ir, (0, 2, 1))))
        # the reference permutes the 4-D [B, C*F, T, 2] tensor with (1, 3, 2, 0) -- giving [C*F, 2, T, B] -- and then
        # reshapes THAT buffer to [B, C, F, T, 2] (UDSS.py:236-238): a scramble of the axes, reproduced as it is
        y = torch.permute(torch.stack((re, im), dim=-1), (1, 3, 2, 0)).reshape(B, C, F, T, 2)
        return stack_complex(y).to(x.dtype)

    def forward(self, x):
        require_device_tensor(x, "UDSS.cRNN")
        return unstack_complex(self.forward_stacked(stack_complex(x)))


class CRM(nn.Module):
    """UDSS.py:91-103."""

    def forward(self, x):
        m = torch.squeeze(torch.tanh(x), 1)
        return torch.complex(m[..., 0], m[..., 1])

    def output(self, X, M):
        return X * M


class CM(nn.Module):
    """UDSS.py:106-116."""

    def forward(self, x):
        m = torch.squeeze(x, 1)
        return torch.complex(m[..., 0], m[..., 1])

    def output(self, X, M):
        return M


class CMEA(nn.Module):
    """UDSS.py:120-177.  ``forward`` takes the head's ``[B, 1, F, T, 2]`` output; on the device the three (3,1)
    convolutions run on the thin-layer kernel, ``output`` is elementwise torch arithmetic."""

    def __init__(self, in_channels=1, mag_f_dim=3):
        super().__init__()
        self.mag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.real_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.imag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.mag_f_dim = mag_f_dim
        self._fused = None

    @staticmethod
    def custom_atan2(y, x):
        pi = torch.tensor([torch.pi], dtype=y.dtype, device=y.device)
        ans = torch.atan(y / (x + 1e-6))
        ans += ((y > 0) & (x < 0)) * pi
        ans -= ((y < 0) & (x < 0)) * pi
        ans *= 1 - ((y > 0) & (x == 0)) * 1.0
        ans += ((y > 0) & (x == 0)) * (pi / 2)
        ans *= 1 - ((y < 0) & (x == 0)) * 1.0
        ans += ((y < 0) & (x == 0)) * (-pi / 2)
        return ans

    def from_parts(self, re, im):
        """re, im: [B, C, F, T] fp32 device tensors."""
        if self._fused is None:
            object.__setattr__(self, "_fused", tuple(FusedConv(m) for m in (self.mag_mask, self.real_mask, self.imag_mask)))
        mag = torch.sqrt(re ** 2 + im ** 2).contiguous()
        if self.training:                  # the three thin convolutions as differentiable nodes
            from . import _lib
            from .train import conv_norm_act
            return tuple(conv_norm_act(m, None, _lib.ACT_NONE, None, t.contiguous()).squeeze(1)
                         for m, t in zip((self.mag_mask, self.real_mask, self.imag_mask), (mag, re, im)))
        return tuple(f(t.contiguous()).squeeze(1) for f, t in zip(self._fused, (mag, re, im)))

    def forward(self, x):
        require_device_tensor(x, "CMEA")
        return self.from_parts(x[..., 0], x[..., 1])

    def output(self, feature, mask, eps=1e-9):
        mag_mask, real_mask, imag_mask = mask
        mag = torch.abs(feature)
        pha = self.custom_atan2(feature.imag, feature.real)
        mag = mag * torch.sigmoid(mag_mask)
        mag_mask = torch.sqrt(torch.clamp(real_mask ** 2 + imag_mask ** 2, eps))
        pha_mask = self.custom_atan2(imag_mask + eps, real_mask + eps)
        real = mag * mag_mask.sigmoid() * torch.cos(pha + pha_mask)
        imag = mag * mag_mask.sigmoid() * torch.sin(pha + pha_mask)
        return torch.stack([real, imag], dim=-1)


def udss_table(model_complexity, input_channels):
    """Layer table of ``UDSS.set_size`` (UDSS.py:582-683)."""
    mc = model_complexity
    return dict(
        enc_channels=[input_channels, mc, mc] + [mc * 2] * 7 + [128],
        enc_kernel_sizes=[(7, 1), (1, 7), (7, 5), (7, 5)] + [(5, 3)] * 6,
        enc_strides=[(1, 1), (1, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1)],
        enc_paddings=[(3, 0), (0, 3), (3, 2), (3, 2)] + [(2, 1)] * 6,
        dec_channels=[0] + [mc * 2] * 11,
        dec_kernel_sizes=[(5, 3)] * 6 + [(7, 5), (7, 5), (1, 7), (7, 1)],
        dec_strides=[(2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (2, 1), (2, 2), (1, 1), (1, 1)],
        dec_paddings=[(2, 1)] * 6 + [(3, 2), (3, 2), (0, 3), (3, 0)],
        dec_output_paddings=[(0, 0), (0, 1)] * 4 + [(0, 0), (0, 0)])


class UDSS(nn.Module):
    """``complex=True``, ``type_encoder="Complex"``, bottleneck "None" / "TGRU" / "FTGRU" / "cRNN", masks "CRM" / "CM" / "CMEA".  As in the
    reference the default ``type_norm`` ("BatcNorm2d", sic) fails at construction; pass "BatchNorm2d"."""

    _announce = "UDSS"          # the name printed with the model complexity (the reference prints at construction)

    def __init__(self, input_channels=4, n_fft=512, complex=True, model_complexity=45, bottleneck="None",
                 padding_mode="zeros", type_encoder="Complex", type_masking="CRM", activation="LeakyReLU",
                 dropout=0.0, type_norm="BatcNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is unet_amd.DSUNet")
        self.bottleneck = bottleneck
        self.complex = complex
        self.n_channel = input_channels
        model_complexity = int(model_complexity // 1.414)
        print("{}::complexity {}".format(self._announce, model_complexity))
        self.set_size(model_complexity=model_complexity, input_channels=input_channels, model_depth=20)
        L = self.model_length = 10
        self.dropout = dropout
        if type_encoder != "Complex":
            raise Exception("Not Implemented")          # ComplexDepthSeparable fails in the reference's constructor
        encs = [Encoder(self.enc_channels[i], self.enc_channels[i + 1], kernel_size=self.enc_kernel_sizes[i],
                        stride=self.enc_strides[i], padding=self.enc_paddings[i], padding_mode=padding_mode,
                        dropout=dropout, activation=activation, type_norm=type_norm) for i in range(L)]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [Decoder(self.dec_channels[i] + self.enc_channels[L - i], self.dec_channels[i + 1],
                        kernel_size=self.dec_kernel_sizes[i], stride=self.dec_strides[i],
                        padding=self.dec_paddings[i], output_padding=self.dec_output_paddings[i],
                        activation=activation) for i in range(L)]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        if bottleneck == "cRNN":
            self.BTN = cRNN(128 * 2)
        elif bottleneck == "TGRU":
            self.BTN = TGRUBlock(128, 256, 128)
        elif bottleneck == "FTGRU":
            self.BTN = FTGRUBlock(128, 256, 128)
        elif bottleneck in ("RNN_AS", "AATT"):
            raise NotImplementedError("UDSS: bottleneck {} is not built".format(bottleneck))
        else:
            self.BTN = nn.Identity()
        self.Attractor = Attractor(n_ch=4, n_fft=n_fft)
        self.attractEncoders = [AttractEncoder(n_ch=self.enc_channels[i + 1]) for i in range(L)]
        for i, m in enumerate(self.attractEncoders):
            self.add_module("attractEncoder_{}".format(i), m)
        self.add_module("linear", ComplexConv2d(self.dec_channels[-1], 1, 1))
        self.padding_mode = padding_mode
        if type_masking == "CRM":
            self.mask = CRM()
        elif type_masking == "CMEA":
            self.mask = CMEA()
        elif type_masking == "CM":
            self.mask = CM()
        else:
            raise Exception("Not Implemented")
        self.dr = nn.Dropout(self.dropout)
        self.decoders = nn.ModuleList(decs)
        self.encoders = nn.ModuleList(encs)
        self.attractors = nn.ModuleList(self.attractEncoders)
        self._head = None
        # A bfloat16 input is widened to fp32 by default.  True: 16-bit operands on the matrix cores over the
        # channel-blocked layout (unet_amd/blocked.py), 4-6x faster.  operand_format "fp16" (default): the input and
        # the first encoder's operands stay bfloat16 (that layer takes two weight halves), everything behind its
        # BatchNorm is fp16.  With "bf16" throughout, bfloat16 activations between the 21 complex layers of this
        # network cost 7e-3 relative L2 error and up to 6e-2 of the output range at single points, whatever the
        # weights' precision (measured against the fp32 CPU forward).
        self.bf16_matrix = False
        self.operand_format = "fp16"
        self.on_fp16_overflow = "defer"       # or "rerun", "nan": unet_amd/blocked.py:fp16_overflow_guard
        # "bf16" format only: 2 = each weight goes in as bfloat16 + its rounding remainder (two MFMAs); with 1 the
        # L2 error is 1.04e-2
        self.bf16_weight_halves = 2

    def set_size(self, model_complexity, model_depth=20, input_channels=1):
        for k, v in udss_table(model_complexity, input_channels).items():
            setattr(self, k, v)

    def forward(self, sf, SV, theta):
        require_device_tensor(sf, "UDSS")
        if sf.dim() != 5 or sf.shape[-1] != 2:
            raise ValueError("UDSS: sf must be [B, C, F, T, 2]")
        if self.training:
            return self._forward_train(sf, SV, theta)
        attract = self.Attractor(SV.float(), theta.float())
        L = self.model_length
        if sf.dtype == torch.bfloat16:
            if self.bf16_matrix and isinstance(self.BTN, nn.Identity) and all(
                    isinstance(m.bn, ComplexBatchNorm2d) and isinstance(m.relu, (nn.LeakyReLU, nn.PReLU))
                    for m in list(self.encoders) + list(self.decoders)):
                from .blocked import fp16_overflow_guard
                return fp16_overflow_guard(self, self._forward_blocked(sf, attract),
                                           lambda: self._forward_blocked(sf, attract))
            sf = sf.float()
        x = stack_complex(sf)
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(x)
            x = enc.forward_stacked(x, self.attractEncoders[i](attract).float().contiguous())
        p = x if isinstance(self.BTN, nn.Identity) else self.BTN.forward_stacked(x)
        for i, dec in enumerate(self.decoders):
            p = dec.forward_stacked(p, None if i == 0 else skips[L - i])
        if self._head is None:
            act = nn.Tanh() if isinstance(self.mask, CRM) else None
            object.__setattr__(self, "_head", _ComplexLayer(self.linear, None, act))
        m = self._head(p).float()                                       # [B, 2, F, T]: re, im (squashed for CRM)
        if isinstance(self.mask, CMEA):
            return self.mask.from_parts(m[:, 0:1], m[:, 1:2])
        return torch.complex(m[:, 0, ...], m[:, 1, ...])

    def _forward_train(self, sf, SV, theta):
        """UDSS.py:561-580 under autograd: every complex block is one differentiable node on the HIP kernels
        (_complex_train); the attractor, its gates, the recurrent bottlenecks and the mask are torch operations."""
        if sf.dtype != torch.float32:
            raise NotImplementedError("UDSS: train mode is fp32 only")
        attract = self.Attractor(SV.float(), theta.float())
        L = self.model_length
        x = stack_complex(sf)
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(x)
            x = enc.forward_stacked(x, self.attractEncoders[i](attract).float())
        p = x if isinstance(self.BTN, nn.Identity) else self.BTN.forward_stacked(x)
        for i, dec in enumerate(self.decoders):
            p = dec.forward_stacked(p, None if i == 0 else skips[L - i])
        m = _complex_train(self.linear, None, nn.Tanh() if isinstance(self.mask, CRM) else None, p)
        if isinstance(self.mask, CMEA):
            return self.mask.from_parts(m[:, 0:1], m[:, 1:2])
        return torch.complex(m[:, 0, ...], m[:, 1, ...])

    def _forward_blocked(self, sf, attract):
        from .blocked import BlockedTensor
        L = self.model_length
        fmt = self.operand_format
        if fmt not in ("fp16", "bf16"):
            raise ValueError("UDSS.operand_format is 'fp16' or 'bf16'")
        halves = self.bf16_weight_halves if fmt == "bf16" else 1           # fp16 weights need no second half
        x = BlockedTensor.from_nchw(stack_complex(sf))
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(x)
            if enc._layer is None:
                object.__setattr__(enc, "_layer", _ComplexLayer(enc.conv, enc.bn, enc.relu))
            x = enc._layer.blocked(x, None, self.attractEncoders[i](attract).float(),
                                   halves=2 if (i == 0 and fmt == "fp16") else halves, out_fmt=fmt)
        for i, dec in enumerate(self.decoders):
            if dec._layer is None:
                object.__setattr__(dec, "_layer", _ComplexLayer(dec.transconv, dec.bn, dec.relu))
            x = dec._layer.blocked(x, None if i == 0 else skips[L - i], halves=halves, out_fmt=fmt)
        if self._head is None:
            act = nn.Tanh() if isinstance(self.mask, CRM) else None
            object.__setattr__(self, "_head", _ComplexLayer(self.linear, None, act))
        head = self._head
        if getattr(head, "_plain", None) is None:
            head._plain = _ComplexLayer(self.linear, None, None)
        m = head._plain.blocked(x, nchw_out=True, halves=halves)     # fp32 [B, 2, F, T]
        if isinstance(self.mask, CRM):
            m = torch.tanh(m)
        if isinstance(self.mask, CMEA):
            return self.mask.from_parts(m[:, 0:1], m[:, 1:2])
        return torch.complex(m[:, 0, ...], m[:, 1, ...])

    def masking(self, X, M):
        return self.mask.output(X, M)


def _helper_base():
    from .DSUNet import DSUNet_helper
    return DSUNet_helper


class UDSS_helper(_helper_base()):
    """Wave in, wave out around the complex-valued network, with the reference's constructor and
    ``forward(x, angle, mic_pos)`` (UDSS.py:685-839).  Transforms, steering vector and delay-and-sum are
    ``DSUNet_helper``'s (same arithmetic in the reference); the spectral feature is ``[B, C, F, T, 2]``."""

    def __init__(self, n_fft=512, complex=True, dropout=0.0, bottleneck="None", model_complexity=45,
                 type_encoder="Complex", use_SV=True, mag_phase=False, corr=False, DSB=False, activation="LeakyReLU",
                 type_norm="BatchNorm2d", type_masking="CRM"):
        nn.Module.__init__(self)
        from .stft import DeviceSTFT
        self.n_fft = n_fft
        self.n_hfft = n_fft // 2 + 1
        self.n_channel = 4
        model_channel = self.n_channel + (1 if DSB else 0)
        self.complex, self.mag_phase, self.use_SV, self.corr, self.DSB = complex, mag_phase, use_SV, corr, DSB
        self.net = UDSS(input_channels=model_channel, complex=complex, dropout=dropout, bottleneck=bottleneck,
                        model_complexity=model_complexity, type_encoder=type_encoder, activation=activation,
                        type_norm=type_norm, type_masking=type_masking)
        self.sr, self.ss, self.dist = 16000, 340.4, 100
        self.window = torch.hann_window(self.n_fft)
        self._stft = DeviceSTFT(n_fft, self.window)

    def forward(self, x, angle, mic_pos):
        require_device_tensor(x, "UDSS_helper")
        B, C, L = x.shape
        X = self._stft.stft(x.reshape(B * C, L).contiguous())
        T = X.shape[2]
        X = torch.nn.functional.pad(X, (0, 16 - T % 16)).reshape(B, C, self.n_hfft, -1)
        feat_src = X
        if self.corr:
            feat_src = torch.cat([X[:, :1], X[:, :1] * X[:, 1:]], dim=1)
        tdoa = self._tdoa(angle, mic_pos, x.device)
        SV = self.steering_vector(angle, mic_pos, x.device, tdoa=tdoa)
        SV = torch.cat((SV.real, SV.imag), -1)
        theta = self.anlge_pre(angle.to(x.device, non_blocking=True))
        feat = feat_src
        if self.DSB:
            feat = torch.cat((feat_src, self.delay_n_sum(feat_src, angle, mic_pos, tdoa=tdoa)[:, None]), dim=1)
        spectral_feature = torch.view_as_real(feat.contiguous())            # [B, C(+1), F, T, 2]
        if self.bf16_matrix:
            self.net.bf16_matrix = True
            spectral_feature = spectral_feature.to(torch.bfloat16)
        mask = self.net(spectral_feature, SV, theta)
        Y = self._apply_mask(feat_src[:, 0], mask)
        return self._stft.istft(Y.contiguous(), L)

    def _apply_mask(self, X0, mask):
        return self.net.masking(X0, mask)
