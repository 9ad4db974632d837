"""Device building blocks with the reference's constructors (UNet_m.py).

``Encoder`` / ``Decoder`` keep the reference's submodule names (``conv`` /
``transconv``, ``bn``, ``acti``, ``dropout``) so checkpoints load unchanged; their
``forward`` is one fused HIP launch (conv or transposed conv, eval-BatchNorm
affine and activation in the epilogue).  ``Decoder.forward`` additionally
accepts the skip tensor, read as extra input channels inside the kernel
instead of through ``torch.cat``.
"""

import torch
import torch.nn as nn

from . import _lib
from .fused import FusedConv, instance_norm_act, layer_norm_act, _pair, require_device_tensor
from .ops import dwconv2d_fwd, resblock_layer


def _activation(name, where):
    # same switch and message as UNet_m.py:324-337
    if name == "LeakyReLU":
        return nn.LeakyReLU(inplace=True)
    if name == "SiLU":
        return nn.SiLU(inplace=True)
    if name == "Softplus":
        return nn.Softplus()
    if name == "PReLU":
        return nn.PReLU()
    if name == "ReLU":
        return nn.ReLU()
    if name == "GELU":
        return nn.GELU()
    raise Exception("ERROR::{}:Unknown activation type {}".format(where, name))


def _norm(name):
    # UNet_m.py:312-319
    if name == "BatchNorm2d":
        return nn.BatchNorm2d
    if name == "InstanceNorm2d":
        return nn.InstanceNorm2d
    if name == "LayerNorm":
        return nn.LayerNorm
    raise Exception("ERROR::Encoder: unknown normalization - {}".format(name))


class _FusedBlock(nn.Module):
    """Shared forward of Encoder / Decoder: fused when the norm is an eval-mode
    BatchNorm2d and the activation is one the epilogue implements; otherwise
    the HIP convolution followed by torch's norm/activation on the device."""

    # 0: fp32 products (exact-fp32 MFMA).  1 / 2: the network this block belongs to asked for bfloat16 OPERANDS on
    # tensors that stay fp32 NCHW (ResUNet on a bfloat16 input): the input is converted to the channel-blocked bfloat16
    # layout on the way in (a summed skip is added in that pass), the convolution runs on the bf16 matrix instruction
    # (16x the fp32 rate) with fp32 accumulation and writes fp32 NCHW, an added tensor included; 2 = weights as
    # bfloat16 plus their rounding remainder.  Blocks the bf16 kernel does not serve keep the fp32 path.
    bf16_operands = 0
    bf16_operand_format = "fp16"      # 16-bit format of those operands: "fp16" or "bf16"

    def _setup(self, conv_attr):
        self._conv_attr = conv_attr
        self._fused = None
        self._blocked = {}

    def _blocked_layers(self, halves):
        """BlockedConv objects covering this block's output channels: one for a layer of at most 128 outputs or a
        multiple of 128; any other width runs as several launches on one converted input, over copies of its parameter
        slices that are rebuilt when a parameter changes.  False when the block is not one that path serves."""
        from .blocked import BlockedConv
        from .fused import activation_code
        conv = getattr(self, self._conv_attr)
        code = activation_code(self.acti)
        K = conv.out_channels
        pointwise = _pair(conv.kernel_size) == (1, 1)
        ok = (conv.groups == 1 and getattr(conv, "padding_mode", "zeros") == "zeros" and not isinstance(conv.padding, str)
              and not (pointwise and conv.in_channels < 128)          # narrow 1x1 layers are bandwidth-bound: fp32 kernel
              and isinstance(self.bn, (nn.BatchNorm2d, nn.Identity))
              and (isinstance(self.bn, nn.Identity) or self.bn.running_mean is not None)
              and code is not None and code[0] in (_lib.ACT_NONE, _lib.ACT_LEAKY, _lib.ACT_RELU)
              and (K <= 128 or K % 128 == 0 or not isinstance(code[1], torch.Tensor) or code[1].numel() == 1))
        if not ok:
            return False
        if K <= 128 or K % 128 == 0:          # one launch (128-channel chunks are a grid dimension of the kernel)
            return [(0, K, BlockedConv(conv, self.bn, self.acti, halves=halves, nchw_out=True))]
        transposed = isinstance(conv, nn.ConvTranspose2d)
        parts = []
        with torch.no_grad():
            for lo in range(0, K, 128):
                hi = min(K, lo + 128)
                kw = dict(kernel_size=conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation,
                          bias=conv.bias is not None)
                if transposed:
                    sub = nn.ConvTranspose2d(conv.in_channels, hi - lo, output_padding=conv.output_padding, **kw)
                    sub.weight.copy_(conv.weight[:, lo:hi])
                else:
                    sub = nn.Conv2d(conv.in_channels, hi - lo, **kw)
                    sub.weight.copy_(conv.weight[lo:hi])
                if conv.bias is not None:
                    sub.bias.copy_(conv.bias[lo:hi])
                norm = self.bn
                if isinstance(norm, nn.BatchNorm2d):
                    norm = nn.BatchNorm2d(hi - lo, eps=self.bn.eps, affine=self.bn.affine).eval()
                    norm.running_mean.copy_(self.bn.running_mean[lo:hi])
                    norm.running_var.copy_(self.bn.running_var[lo:hi])
                    if self.bn.affine:
                        norm.weight.copy_(self.bn.weight[lo:hi])
                        norm.bias.copy_(self.bn.bias[lo:hi])
                sub, norm = sub.to(conv.weight.device), norm.to(conv.weight.device)
                parts.append((lo, hi, BlockedConv(sub, norm, self.acti, halves=halves, nchw_out=True)))
        return parts

    def _run_bf16_operands(self, x, skip, in_mode, residual, halves):
        """The block through unet_amd.blocked on fp32 NCHW tensors, or None when it is not one that path serves."""
        from .blocked import BlockedTensor
        conv = getattr(self, self._conv_attr)
        srcs = [conv.weight, conv.bias] + ([self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
                                           if isinstance(self.bn, nn.BatchNorm2d) else [])
        key = (halves, isinstance(self.bn, nn.BatchNorm2d) and self.bn.training) + tuple(
            None if t is None else (t.data_ptr(), t._version) for t in srcs)
        hit = self._blocked.get(halves)
        if hit is None or (hit[0] != key and hit[1] is not False and len(hit[1]) > 1) or (hit[0] != key and hit[1] is False):
            # single-launch layers track parameter changes themselves (BlockedConv._versions); sliced copies and a
            # refusal are redone when a parameter (or the norm's mode) changes
            hit = self._blocked[halves] = (key, self._blocked_layers(halves))
        parts = hit[1]
        if parts is False:
            return None
        try:
            # fp16 operands: these layers sit behind BatchNorms (modest range), and fp16 keeps three more bits
            if skip is not None and in_mode == _lib.IN_ADD:
                xb, sb = BlockedTensor.from_nchw(x.contiguous(), addend=skip.contiguous(), fmt=self.bf16_operand_format), None
            else:
                xb = BlockedTensor.from_nchw(x.contiguous(), fmt=self.bf16_operand_format)
                sb = None if skip is None else BlockedTensor.from_nchw(skip.contiguous(), fmt=self.bf16_operand_format)
            outs = [layer(xb, sb, None, residual=None if residual is None else residual[:, lo:hi].contiguous())
                    for lo, hi, layer in parts]
            return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
        except _lib.UnetmiError as e:
            if e.code not in (_lib.E_UNSUPPORTED, _lib.E_BAD_DESC):
                raise
            self._blocked[halves] = (key, False)    # e.g. LDS images of this geometry exceed a CU: fp32 path from now on
            return None

    def _layer(self):
        if self._fused is None:
            object.__setattr__(self, "_fused",
                               FusedConv(getattr(self, self._conv_attr), self.bn, self.acti))
        return self._fused

    def _run(self, x, skip=None, in_mode=None, gate=None, residual=None):
        require_device_tensor(x, type(self).__name__)
        if self.training:
            return self._run_differentiable(x, skip, in_mode, gate, residual)
        if self.bf16_operands and x.dtype == torch.float32 and gate is None:
            y = self._run_bf16_operands(x, skip, in_mode, residual, int(self.bf16_operands))
            if y is not None:
                return y
        layer = self._layer()
        if layer.fusable():
            return layer(x, skip, in_mode=in_mode, gate=gate, residual=residual)
        y = layer(x, skip, in_mode=in_mode, epilogue=False)
        fused = instance_norm_act(y, self.bn, self.acti, gate)       # InstanceNorm2d + activation (+ gate): one launch
        if fused is None:
            fused = layer_norm_act(y, self.bn, self.acti, gate)      # LayerNorm over the time axis, likewise
        if fused is not None:
            return fused if residual is None else fused + residual
        y = self.acti(self.bn(y))
        if gate is not None:
            y = y * gate[:, :, None, None]
        if residual is not None:
            y = y + residual
        return y

    def _run_differentiable(self, x, skip, in_mode, gate, residual):
        """train() mode: the convolution, the BatchNorm (batch statistics) and an activation the epilogue knows by
        code are HIP launches recorded as one autograd node (unet_amd/train.py: forward, input gradient and weight
        gradient on the library's kernels).  What that node does not cover stays a torch operation on the device, so
        that autograd differentiates it: an activation with a learnable parameter (PReLU), another norm
        (InstanceNorm2d / LayerNorm), dropout between convolution and norm, a summed skip, a gate, an added tensor.
        eval() mode is inference only: its output carries no autograd history."""
        from .fused import activation_code
        from .train import batch_norm_act, conv_norm_act
        conv = getattr(self, self._conv_attr)
        found = activation_code(self.acti)
        by_code = found is not None and not isinstance(found[1], torch.Tensor)
        code, par = found if by_code else (_lib.ACT_NONE, None)
        act_param = None if par is None else torch.full((1,), par, dtype=torch.float32, device=x.device)
        if skip is not None and in_mode == _lib.IN_ADD:
            x, skip = x + skip, None
        drop = getattr(self, "dropout", None)
        dropping = drop is not None and drop.p != 0.0
        plain_norm = isinstance(self.bn, (nn.BatchNorm2d, nn.Identity))
        if plain_norm and not dropping:
            y = conv_norm_act(conv, self.bn, code, act_param, x.contiguous(), skip)
        else:
            z = conv_norm_act(conv, None, _lib.ACT_NONE, None, x.contiguous(), skip)
            if dropping:
                z = drop(z)
            if isinstance(self.bn, nn.BatchNorm2d):
                y = batch_norm_act(self.bn, z, code, act_param)
            else:
                y = self.bn(z)
                by_code = False                         # the activation follows as a torch operation
        if not by_code:
            y = self.acti(y)
        if gate is not None:
            y = y * gate[:, :, None, None]
        if residual is not None:
            y = y + residual
        return y


class Encoder(_FusedBlock):
    """Conv2d -> Dropout -> norm -> activation (UNet_m.py:304-345)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None, dilation=1,
                 norm="BatchNorm2d", padding_mode="zeros", dropout=0.0, activation="LeakyReLU"):
        super().__init__()
        if padding is None:
            padding = [(k - 1) // 2 for k in kernel_size]
        norm_cls = _norm(norm)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, padding_mode=padding_mode)
        self.bn = norm_cls(out_channels)
        self.acti = _activation(activation, "Encoder")
        self.dropout = nn.Dropout(dropout)
        self._setup("conv")

    def forward(self, x, gate=None):
        return self._run(x, gate=gate)


class Decoder(_FusedBlock):
    """ConvTranspose2d -> norm -> activation (UNet_m.py:347-383).

    ``forward(x, skip)`` computes ``decoder(torch.cat([x, skip], 1))`` without
    forming the concatenation; ``forward(x, skip, add=True)`` computes
    ``decoder(x + skip)``; ``residual=`` adds a tensor to the output."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=(0, 0), dilation=1,
                 output_padding=(0, 0), activation="LeakyReLU", norm="BatchNorm2d"):
        super().__init__()
        norm_cls = _norm(norm)
        self.transconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size,
                                            stride=stride, dilation=dilation,
                                            output_padding=output_padding, padding=padding)
        self.bn = norm_cls(out_channels)
        self.acti = _activation(activation, "Encoder")
        self._setup("transconv")

    def forward(self, x, skip=None, add=False, residual=None):
        """``residual`` (shaped like the output) is added to the block's output in the kernel's epilogue: the
        ``y + res`` in front of the NEXT block of an additive-skip decoder (ResUNet.py:209), taken here so that
        block reads one tensor."""
        return self._run(x, skip, in_mode=(_lib.IN_ADD if add else None) if skip is not None else None,
                         residual=residual)


class ResBlock(nn.Module):
    """Five [depthwise 3x3 dilated along time -> pointwise 1x1 -> PReLU -> BatchNorm]
    layers, the first replacing x and the others adding to the running value
    (UNet_m.py:518-563).  One PReLU instance is shared by the five layers, as in the
    reference, so ``res_{0..4}.2.weight`` alias one tensor.

    With 64 channels (every block of ResUNet) a layer is ONE launch that reads the activation
    once and writes it once (unetmi_resblock_layer_fwd).  Other widths take two launches per
    layer: the depthwise conv, then pointwise conv + bias + PReLU + BatchNorm affine + the
    running-value add.  ``forward(x, add)`` starts from ``x + add`` without forming it (the
    skip add in front of ResUNet's decoder blocks, ResUNet.py:209)."""

    TIME_DILATIONS = (1, 2, 4, 8, 16)

    def __init__(self, n_channel, norm="BatchNorm2d", dropout=0.0, activation="PReLU"):
        super().__init__()
        if norm != "BatchNorm2d":
            raise Exception("ERROR::ResBlock: unknown nromalization - {}".format(norm))
        if activation != "PReLU":
            # the reference's GELU branch never defines the local it then uses (UNet_m.py:530-551)
            raise Exception("ERROR::ResBlock:Unknown activation type " + str(activation))
        shared = nn.PReLU()
        self.layers = []
        for i, d in enumerate(self.TIME_DILATIONS):
            layer = nn.Sequential(
                nn.Conv2d(n_channel, n_channel, (3, 3), (1, 1), padding=(1, d), dilation=(1, d), groups=n_channel),
                nn.Conv2d(n_channel, n_channel, (1, 1)),
                shared,
                nn.BatchNorm2d(n_channel))
            self.add_module("res_{}".format(i), layer)
            self.layers.append(layer)
        self._fused = None

    def forward(self, x, add=None):
        require_device_tensor(x, "ResBlock")
        if self.training:
            return self._forward_train(x, add)
        if self._fused is None:
            object.__setattr__(self, "_fused", [FusedConv(l[1], l[3], l[2], act_first=True) for l in self.layers])
        if x.shape[1] == 64 and self.layers[0][2].weight.numel() == 1:
            return self._forward_one_launch_per_layer(x, add)
        r = None
        for i, layer in enumerate(self.layers):
            dw = layer[0]
            src = x if i == 0 else r
            t = dwconv2d_fwd(src, add if i == 0 else None, dw.weight.float(), None if dw.bias is None else dw.bias.float(),
                             list(_pair(dw.padding)), list(_pair(dw.dilation)))
            r = self._fused[i](t, residual=r)        # layer 0 has no running value to add
        return r

    def _forward_train(self, x, add):
        """train() mode under autograd: per layer the depthwise convolution (unetmi::dwconv2d_fwd; its input gradient
        is the same kernel with reversed taps, its weight gradient unetmi_dwconv2d_wgrad), the pointwise convolution
        (unetmi::conv_norm_act without norm: forward, input and weight gradient on the MFMA kernels), the shared PReLU
        (a torch operation, so its slope gets a gradient) and BatchNorm with batch statistics (unetmi::bn_act)."""
        from .train import batch_norm_act, conv_norm_act
        r = None
        for i, layer in enumerate(self.layers):
            dw, pw, act, bn = layer[0], layer[1], layer[2], layer[3]
            src = (x if i == 0 else r).contiguous()
            t = dwconv2d_fwd(src, add.contiguous() if (i == 0 and add is not None) else None, dw.weight, dw.bias,
                             list(_pair(dw.padding)), list(_pair(dw.dilation)))
            y = batch_norm_act(bn, act(conv_norm_act(pw, None, _lib.ACT_NONE, None, t)))
            r = y if i == 0 else r + y
        return r

    def _forward_one_launch_per_layer(self, x, add):
        r = None
        for i, layer in enumerate(self.layers):
            dw, pw = layer[0], layer[1]
            st = self._fused[i]._prepare(x.device)      # folded BatchNorm scale/shift, PReLU parameter on device
            f32 = lambda t: None if t is None else (t if t.dtype == torch.float32 else t.float())  # noqa: E731
            r = resblock_layer(x if i == 0 else r, add if i == 0 else None, f32(dw.weight), f32(dw.bias), f32(pw.weight),
                               f32(pw.bias), st["act_param"], st["scale"], st["shift"], int(_pair(dw.dilation)[1]), i > 0)
        return r


class DepthwiseSeparableConv2d(nn.Module):
    """UNet_m.py:7-31 (parameter names as there; torch's device ops -- only the multi-scale side branch of
    ``ResUNetOnFreq2`` uses it, and nothing reads that branch's result)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   groups=in_channels)
        self.bn_depth = nn.BatchNorm2d(out_channels)
        self.pointwise = nn.Conv2d(out_channels, out_channels, kernel_size=1)
        self.bn_point = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        require_device_tensor(x, "DepthwiseSeparableConv2d")
        x = torch.relu(self.bn_depth(self.depthwise(x)))
        return torch.relu(self.bn_point(self.pointwise(x)))


class GlobalChannelLayerNorm(nn.Module):
    """UNet_m.py:739-763: one mean / variance per sample, per-channel gain and offset."""

    def __init__(self, channel_size):
        super().__init__()
        self.channel_size = channel_size
        self.gamma = nn.Parameter(torch.ones(channel_size))
        self.beta = nn.Parameter(torch.zeros(channel_size))

    def forward(self, x):
        require_device_tensor(x, "GlobalChannelLayerNorm")
        dims = list(range(1, x.dim()))
        mean = x.mean(dim=dims, keepdim=True)
        var = torch.pow(x - mean, 2).mean(dim=dims, keepdim=True)
        normed = (x - mean) / (var + 1e-8).sqrt()
        return (self.gamma * normed.transpose(1, -1) + self.beta).transpose(1, -1)


class MultiScaleConvBlock(nn.Module):
    """UNet_m.py:771-788."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.net = nn.Sequential(nn.Conv2d(c, 2 * c, kernel_size=1), GlobalChannelLayerNorm(2 * c),
                                 DepthwiseSeparableConv2d(2 * c, 2 * c, kernel_size=1), GlobalChannelLayerNorm(2 * c),
                                 nn.Conv2d(2 * c, c, kernel_size=1), GlobalChannelLayerNorm(c))

    def forward(self, x):
        return self.net(x)


def custom_atan2(y, x):
    """UNet_m.py:677-686."""
    pi = torch.tensor([torch.pi], dtype=y.dtype, device=y.device)
    ans = torch.atan(y / (x + 1e-6))
    ans += ((y > 0) & (x < 0)) * pi
    ans -= ((y < 0) & (x < 0)) * pi
    ans *= 1 - ((y > 0) & (x == 0)) * 1.0
    ans += ((y > 0) & (x == 0)) * (pi / 2)
    ans *= 1 - ((y < 0) & (x == 0)) * 1.0
    ans += ((y < 0) & (x == 0)) * (-pi / 2)
    return ans


class MEA(nn.Module):
    """Mask estimation and application head (UNet_m.py:691-732).  The three (3,1) convolutions run on the
    thin-layer kernel; ``output`` is elementwise torch arithmetic on the device (the reference's own
    ``output`` raises NameError -- numpy is used there without being imported -- so it follows the arithmetic
    as written)."""

    def __init__(self, in_channels=4, mag_f_dim=3):
        super().__init__()
        self.mag_mask = nn.Conv2d(in_channels, mag_f_dim, kernel_size=(3, 1), padding=(1, 0))
        self.real_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.imag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.register_buffer("kernel", torch.eye(mag_f_dim).reshape(mag_f_dim, 1, mag_f_dim, 1))
        self.mag_f_dim = mag_f_dim
        self._fused = None

    def forward(self, x):
        require_device_tensor(x, "MEA")
        if self._fused is None:
            object.__setattr__(self, "_fused",
                               tuple(FusedConv(m) for m in (self.mag_mask, self.real_mask, self.imag_mask)))
        x = x.float().contiguous()
        mag, re, im = (f(x) for f in self._fused)
        return (mag, re.squeeze(1), im.squeeze(1))

    def output(self, mask, feature, eps=1e-9):
        mag_mask, real_mask, imag_mask = mask[0], mask[1], mask[2]
        mag = torch.norm(feature, dim=-1)
        pha = custom_atan2(feature[..., 1], feature[..., 0])
        h = (self.mag_f_dim - 1) // 2
        mag = nn.functional.conv2d(nn.functional.pad(mag[:, None], [0, 0, h, h]), self.kernel)
        mag = (mag * mag_mask.relu()).sum(dim=1)
        mag_mask = torch.sqrt(torch.clamp(real_mask ** 2 + imag_mask ** 2, eps))
        pha_mask = custom_atan2(imag_mask + eps, real_mask + eps)
        real = mag * mag_mask.relu() * torch.cos(pha + pha_mask)
        imag = mag * mag_mask.relu() * torch.sin(pha + pha_mask)
        return torch.stack([real, imag], dim=-1)
