"""Device ``UNet`` with the reference's constructor and ``forward`` (UNet.py:6-113).

The concat-skip shell is the reference's; what changes is that a decoder built
from ``unet_amd.UNet_m.Decoder`` receives its skip tensor as a second kernel
input (no ``torch.cat``), and the 1x1 ``linear`` + Sigmoid head is one fused
launch.  Blocks of any other type are called as the reference calls them.
"""

import torch
import torch.nn as nn

from .UNet_m import Decoder, Encoder
from .fused import FusedConv, require_device_tensor


def register_concat_shell(net, encoders, decoders, bottlenecks, res_paths):
    """Submodule registration common to UNet and cRFUNet.UNet20: names
    ``encoder_i`` / ``decoder_i`` / ``res_path{i}`` / ``bottleneck{i}``
    (UNet.py:28-66, cRFUNet.py:229-267).  The res-path list has one trailing
    identity whose output is never consumed."""
    n = len(encoders)
    net.len_model = n
    net.encoders, net.decoders = encoders, decoders
    for i, m in enumerate(encoders):
        net.add_module("encoder_{}".format(i), m)
    for i, m in enumerate(decoders):
        net.add_module("decoder_{}".format(i), m)
    if res_paths is not None:
        if len(res_paths) != n - 1:
            raise Exception("ERROR::unmatched res_path : {} != {}".format(len(res_paths), n - 1))
        paths = list(res_paths)
    else:
        paths = [nn.Identity() for _ in range(n - 1)]
    paths.append(nn.Identity())
    net.res_paths = paths
    for i, m in enumerate(paths):
        net.add_module("res_path{}".format(i), m)
    net.bottlenecks = list(bottlenecks) if bottlenecks is not None else [nn.Identity()]
    for i, m in enumerate(net.bottlenecks):
        net.add_module("bottleneck{}".format(i), m)


def concat_shell_forward(net, x):
    """Encoders, bottleneck, decoders with [decoder output, skip] channel order
    (UNet.py:85-106).  Returns the last decoder's output."""
    skips = []
    for i, enc in enumerate(net.encoders):
        skips.append(net.res_paths[i](x))
        x = enc(x)
        if net.print_shape:
            print("Encoder {} : {}".format(i, x.shape))
    p = x
    for i, b in enumerate(net.bottlenecks):
        p = b(p)
        if net.print_shape:
            print("bottleneck {} : {}".format(i, p.shape))
    last = net.len_model - 1
    skip = None
    for i, dec in enumerate(net.decoders):
        if skip is None:
            p = dec(p)
        elif isinstance(dec, Decoder):
            p = dec(p, skip)                      # concat fused into the kernel's read
        else:
            p = dec(torch.cat([p, skip], dim=1))
        if net.print_shape:
            print("Decoder {} : {}".format(i, p.shape))
        if i == last:
            break
        skip = skips[last - i]
    return p


def blocked_shell_layers(net):
    """BlockedConv per encoder / decoder / head for the bfloat16-operand path (unet_amd/blocked.py), or False when a
    layer is outside what that kernel serves: Encoder / Decoder blocks with eval BatchNorm2d and LeakyReLU / PReLU /
    ReLU, at most 128 output channels, identity res-paths and bottlenecks."""
    fmt = getattr(net, "operand_format", "fp16")
    if fmt not in ("fp16", "bf16"):
        raise ValueError("operand_format is 'fp16' or 'bf16'")
    cached = getattr(net, "_blocked", None)
    if cached is not None and cached[0] == fmt:
        return cached[1]
    from . import _lib
    from .blocked import BlockedConv
    from .fused import activation_code
    ok = all(isinstance(m, nn.Identity) for m in list(net.res_paths) + list(net.bottlenecks))
    for m in list(net.encoders) + list(net.decoders):
        if not isinstance(m, (Encoder, Decoder)):
            ok = False
            break
        conv = m.conv if isinstance(m, Encoder) else m.transconv
        code = activation_code(m.acti)
        ok = ok and isinstance(m.bn, nn.BatchNorm2d) and code is not None and \
            code[0] in (_lib.ACT_NONE, _lib.ACT_LEAKY, _lib.ACT_RELU) and conv.out_channels <= 128 and \
            conv.groups == 1 and getattr(conv, "padding_mode", "zeros") == "zeros"
    built = False
    if ok and net.linear.out_channels <= 128:
        # The network input (an STFT of unknown scale) and the first encoder's operands stay bfloat16 -- that layer
        # takes its weights as two bfloat16 halves, it is bound by its activation traffic anyway -- and with
        # operand_format "fp16" (default) every tensor behind its BatchNorm is stored, and multiplied, as fp16.
        # A one-channel 1x1 head on a one-channel last decoder is a scalar affine map: that decoder then writes plain
        # fp32 [B, 1, F, T] from its accumulators (4 bytes per pixel instead of a 16-byte record that a further launch
        # would read back) and the head is applied elementwise.
        lin = net.linear
        scalar_head = lin.in_channels == 1 and lin.out_channels == 1 and lin.kernel_size == (1, 1) and \
            lin.stride == (1, 1) and lin.padding == (0, 0) and net.decoders[-1].transconv.out_channels == 1
        n_dec = len(net.decoders)
        built = dict(enc=[BlockedConv(m.conv, m.bn, m.acti, halves=2 if i == 0 else 1, out_fmt=fmt)
                          for i, m in enumerate(net.encoders)],
                     dec=[BlockedConv(m.transconv, m.bn, m.acti, out_fmt=fmt,
                                      nchw_out=scalar_head and i == n_dec - 1) for i, m in enumerate(net.decoders)],
                     head=None if scalar_head else BlockedConv(lin, None, None, nchw_out=True))
        # A head of at most 32 outputs takes the network's mask activation in its epilogue (the kernels with the
        # transcendental activations are built for one 32-channel tile): one pass less over the fp32 mask.
        mask_act = getattr(net, "activation_mask", None)
        built["head_has_mask"] = False
        if not scalar_head and lin.out_channels <= 32 and mask_act is not None:
            code = activation_code(mask_act)
            if code is not None and code[0] not in (_lib.ACT_NONE, _lib.ACT_LEAKY, _lib.ACT_RELU):
                built["head"] = BlockedConv(lin, None, mask_act, nchw_out=True)
                built["head_has_mask"] = True
    object.__setattr__(net, "_blocked", (fmt, built))
    return built


def blocked_shell_forward(net, layers, x):
    """The concat shell on channel-blocked 16-bit tensors; returns the head's fp32 [B, K, F, T] output before the mask
    activation -- or after it when ``layers["head_has_mask"]`` says the head's epilogue applied it.  (Fusing UNet's 1x1 head into the last decoder, as DSUNet does, was measured: the one-column-tile
    kernel a fused head needs makes the batch-64 forward 7.5 ms instead of 7.3, at the same error.)"""
    from .blocked import BlockedTensor
    t = BlockedTensor.from_nchw(x.contiguous())
    skips = []
    for enc in layers["enc"]:
        skips.append(t)
        t = enc(t)
    last = net.len_model - 1
    skip = None
    for i, dec in enumerate(layers["dec"]):
        t = dec(t, skip)
        if i == last:
            break
        skip = skips[last - i]
    if layers["head"] is None:
        lin = net.linear
        return t.mul_(lin.weight.detach().reshape(())).add_(lin.bias.detach().reshape(()) if lin.bias is not None else 0.0)
    return layers["head"](t)


class UNet(nn.Module):
    """Drop-in for the reference ``UNet``.  ``print_shape`` is a keyword here; the
    reference reads an undefined global of that name (UNet.py:24)."""

    def __init__(self, encoders, decoders, bottlenecks=None, res_paths=None, device="cuda:0",
                 print_shape=False):
        super().__init__()
        self.device = device
        if len(encoders) != len(decoders):
            raise Exception("ERROR::unmatched legnth : enc {} != dec {}".format(
                len(encoders), len(decoders)))
        self.print_shape = print_shape
        register_concat_shell(self, encoders, decoders, bottlenecks, res_paths)
        self.add_module("linear", nn.Conv2d(1, 1, 1))
        self.activation_mask = nn.Sigmoid()
        self._head = None

    # bfloat16 inputs: 16-bit operands on the matrix cores over the channel-blocked layout when every layer is one that
    # kernel serves (False: the NCHW bfloat16-storage kernels with fp32 products).  operand_format "fp16" (default):
    # the input and the first encoder's operands stay bfloat16, every tensor behind its BatchNorm is fp16.  At the
    # benchmark's size (batch 64, T=509) against the fp32 CPU forward: 7.2 ms, relative L2 2.3e-3, max-norm 7.8e-3;
    # with "bf16" operands throughout 7.0 ms, 7.2e-3 / 4.2e-2; the storage-only path 46 ms, 5.4e-3 / 3.1e-2; the
    # fp32 forward 35.2 ms.
    bf16_matrix = True
    operand_format = "fp16"
    on_fp16_overflow = "defer"        # or "rerun", "nan": unet_amd/blocked.py:fp16_overflow_guard

    def forward(self, x):
        require_device_tensor(x, "UNet")
        if x.dtype == torch.bfloat16 and self.bf16_matrix and not self.training:
            layers = blocked_shell_layers(self)
            if layers:
                def run():
                    ls = blocked_shell_layers(self)
                    p = blocked_shell_forward(self, ls, x)
                    return (p if ls["head_has_mask"] else self.activation_mask(p)).to(x.dtype)
                from .blocked import fp16_overflow_guard
                return fp16_overflow_guard(self, run(), run)
        p = concat_shell_forward(self, x)
        if self.training:
            from . import _lib
            from .train import conv_norm_act
            return conv_norm_act(self.linear, None, _lib.ACT_SIGMOID, None, p)
        if self._head is None:
            object.__setattr__(self, "_head", FusedConv(self.linear, None, self.activation_mask))
        return self._head(p)


def default_unet_blocks(c_in=2, c_last=1, f_dim=32):
    """(encoders, decoders) of the "UNet default config": the 10+10 layer table
    that cRFUNet.py:192-218 inlines, with c_in input channels and a c_last-channel
    last decoder (UNet.py's head is Conv2d(1,1,1), so c_last = 1)."""
    from .tables import unet20_layers
    enc_rows, dec_rows = unet20_layers(c_in=c_in, c_last=c_last, f_dim=f_dim)
    return ([Encoder(*r["args"], **r["kw"]) for r in enc_rows],
            [Decoder(*r["args"], **r["kw"]) for r in dec_rows])
