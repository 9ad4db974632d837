"""Device ``DSUNet`` (directional separating U-Net) with the reference's constructor
and ``forward(sf, SV, theta)`` (DSUNet.py:467-631).

Every encoder is one fused launch: conv, eval-BatchNorm, activation AND the
per-(batch, channel) gate ``max(sigmoid(Linear(code)), 1/C)`` that the reference
multiplies in afterwards (DSUNet.py:599-602).  Decoders read ``[p, skip]`` as two
kernel inputs.  The 1x1 ``linear`` head with the complex-ratio-mask tanh is one
more fused launch.  The attractor MLP (0.8 MFLOP per utterance) runs as ordinary
torch device ops.  Supported variants: type_encoder "Encoder" / "GConv" (gated
convolutions: both convolutions of a layer in one double-width launch, then one gate
pass), bottleneck "None" / "TGRU", type_masking CRM / CRM_sig / CRM_soft / CM.
type_residual "SE" is refused: the reference builds its SE blocks on the wrong channel
count and its forward raises (DSUNet.py:541-545, :588).
"""

import torch
import torch.nn as nn

from . import _lib
from . import gru as _gru
from .fused import FusedConv, activation_code, instance_norm_act, require_device_tensor
from .ops import glu


def _activation(name):
    # DSUNet.py:79-86; unknown names leave the attribute unset there, KeyError here
    return {"LeakyReLU": lambda: nn.LeakyReLU(inplace=True), "PReLU": nn.PReLU, "SiLU": nn.SiLU,
            "Softplus": nn.Softplus}[name]()


def _norm(type_norm):
    return {"BatchNorm2d": nn.BatchNorm2d, "InstanceNorm2d": nn.InstanceNorm2d}[type_norm]


class _Block(nn.Module):
    def _run(self, conv, x, skip=None, gate=None):
        require_device_tensor(x, type(self).__name__)
        if self.training:
            return self._run_differentiable(conv, x, skip, gate)
        if self._fused is None:
            object.__setattr__(self, "_fused", FusedConv(conv, self.bn, self.relu))
        if self._fused.fusable():
            return self._fused(x, skip, gate=gate)
        z = self._fused(x, skip, epilogue=False)
        y = instance_norm_act(z, self.bn, self.relu, gate)             # InstanceNorm2d + activation + gate: one launch
        if y is not None:
            return y
        y = self.relu(self.bn(z))
        return y if gate is None else y * gate[:, :, None, None]


    def _run_differentiable(self, conv, x, skip, gate):
        """train() mode (the reference's blocks are trained under stock autograd, DSUNet.py:62-93, :174-200):
        convolution, BatchNorm with batch statistics and an activation known by code are one autograd node whose
        forward and backward are HIP launches (unet_amd/train.py); the gate product stays a torch operation, so its
        gradient reaches the attractor's Linears.  PReLU (a learnable slope), InstanceNorm2d and dropout between
        convolution and norm are torch operations around the convolution node."""
        from .train import batch_norm_act, conv_norm_act
        found = activation_code(self.relu)
        by_code = found is not None and not isinstance(found[1], torch.Tensor)
        code, par = found if by_code else (_lib.ACT_NONE, None)
        act_param = None if par is None else torch.full((1,), par, dtype=torch.float32, device=x.device)
        drop = getattr(self, "dr", None)
        dropping = drop is not None and drop.p != 0.0
        if isinstance(self.bn, nn.BatchNorm2d) and not dropping:
            y = conv_norm_act(conv, self.bn, code, act_param, x.contiguous(), skip)
        else:
            z = conv_norm_act(conv, None, _lib.ACT_NONE, None, x.contiguous(), skip)
            if dropping:
                z = drop(z)
            if isinstance(self.bn, nn.BatchNorm2d):
                y = batch_norm_act(self.bn, z, code, act_param)
            else:
                y, by_code = self.bn(z), False
        if not by_code:
            y = self.relu(y)
        return y if gate is None else y * gate[:, :, None, None]


class Encoder(_Block):
    """conv -> dropout -> norm -> act (DSUNet.py:62-93)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None, padding_mode="zeros",
                 activation="LeakyReLU", dropout=0.0, type_norm="BatchNorm2d"):
        super().__init__()
        if padding is None:
            padding = [(k - 1) // 2 for k in kernel_size]
        self.dr = nn.Dropout(dropout)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, padding_mode=padding_mode)
        self.bn = _norm(type_norm)(out_channels)
        self.relu = _activation(activation)
        self._fused = None

    def forward(self, x, gate=None):
        return self._run(self.conv, x, gate=gate)


class Decoder(_Block):
    """tconv -> norm -> act; ``output_padding`` precedes ``padding`` (DSUNet.py:174-200)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, output_padding, padding=(0, 0),
                 activation="LeakyReLU", type_norm="BatchNorm2d"):
        super().__init__()
        self.transconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                            output_padding=output_padding, padding=padding)
        self.bn = _norm(type_norm)(out_channels)
        self.relu = _activation(activation)
        self._fused = None

    def forward(self, x, skip=None):
        return self._run(self.transconv, x, skip)


def _gconv_norm(type_norm, channels):
    # DSUNet.py:144-149
    if type_norm == "BatchNorm2d":
        return nn.BatchNorm2d(channels)
    if type_norm == "InstanceNorm2d":
        return nn.InstanceNorm2d(channels, track_running_stats=True)
    return nn.Identity()


def _gconv_activation(name):
    # DSUNet.py:151-162
    return {"ReLU": nn.ReLU, "SiLU": nn.SiLU, "ELU": nn.ELU, "Softplus": nn.Softplus, "PReLU": nn.PReLU,
            "LeakyReLU": nn.LeakyReLU}[name]()


class _Gated(nn.Module):
    """sigmoid(conv_m(x)) * conv_c(x): the two convolutions share input and geometry, so they run as
    one launch of a convolution with the weights stacked to 2K output channels (rebuilt when a
    parameter changes), followed by one gate pass (unetmi_glu_fwd) that also carries the layer's
    BatchNorm affine, activation and DSUNet's attractor gate."""

    def _stacked(self, c, m):
        key = tuple((t._version, t.data_ptr()) for t in (c.weight, c.bias, m.weight, m.bias))
        if self._twin is None or self._twin_key != key:
            transposed = isinstance(c, nn.ConvTranspose2d)
            kw = dict(kernel_size=c.kernel_size, stride=c.stride, padding=c.padding)
            if transposed:
                twin = nn.ConvTranspose2d(c.in_channels, 2 * c.out_channels, output_padding=c.output_padding, **kw)
            else:
                twin = nn.Conv2d(c.in_channels, 2 * c.out_channels, **kw)
            with torch.no_grad():
                twin = twin.to(c.weight.device)
                twin.weight.copy_(torch.cat([c.weight, m.weight], dim=1 if transposed else 0))
                twin.bias.copy_(torch.cat([c.bias, m.bias]))
            object.__setattr__(self, "_twin", FusedConv(twin))
            object.__setattr__(self, "_twin_key", key)
        return self._twin

    def _gated(self, x, skip, norm, act, gate):
        require_device_tensor(x, type(self).__name__)
        c, m = self.c, self.m[0]
        if c.groups != 1:
            raise NotImplementedError("grouped gated convolutions are not built")
        if self.training:
            return self._gated_train(c, m, x, skip, norm, act, gate)
        z = self._stacked(c, m)(x, skip)
        scale = shift = None
        post_norm = None
        if isinstance(norm, nn.BatchNorm2d):
            inv = torch.rsqrt(norm.running_var.float() + norm.eps)
            scale = (norm.weight.float() * inv).contiguous()
            shift = (norm.bias.float() - norm.running_mean.float() * scale).contiguous()
        elif norm is not None and not isinstance(norm, nn.Identity):
            post_norm = norm
        found = activation_code(act) if post_norm is None else None
        if found is None or (isinstance(found[1], torch.Tensor) and found[1].numel() != 1):
            y = glu(z, scale, shift, 0, None, None)                  # gate product (+ BatchNorm) on the device kernel
            if post_norm is not None:
                y = post_norm(y)
            y = act(y) if act is not None else y
            return y if gate is None else y * gate[:, :, None, None]
        code, par = found
        if isinstance(par, torch.Tensor):
            par = par.detach().float().reshape(1)
        elif par is not None:
            par = torch.full((1,), par, dtype=torch.float32, device=x.device)
        return glu(z, scale, shift, code, par, gate)


    def _gated_train(self, c, m, x, skip, norm, act, gate):
        """train() mode: the two convolutions are ONE differentiable node on the stacked weight (assembled from the
        parameters by torch.cat, so its gradient splits back onto them); the sigmoid gate product, a norm other
        than BatchNorm2d and the activation are torch operations; BatchNorm2d runs the library's kernels with batch
        statistics."""
        from .train import batch_norm_act
        transposed = isinstance(c, nn.ConvTranspose2d)
        K = c.out_channels
        W = torch.cat([c.weight, m.weight], dim=1 if transposed else 0).contiguous()
        zero = torch.zeros(K, device=W.device)
        bias = torch.cat([c.bias if c.bias is not None else zero, m.bias if m.bias is not None else zero])
        geom, _ = FusedConv(c).geometry()
        geom = [2 * K] + list(geom[1:])
        z = torch.ops.unetmi.conv_norm_act(x.contiguous(), None if skip is None else skip.contiguous(), W, bias, None,
                                           None, None, None, geom, transposed, 0, 0.0, _lib.ACT_NONE, None)[0]
        y = z[:, :K] * torch.sigmoid(z[:, K:])
        if isinstance(norm, nn.BatchNorm2d):
            y = batch_norm_act(norm, y.contiguous())
        elif norm is not None:
            y = norm(y)
        if act is not None:
            y = act(y)
        return y if gate is None else y * gate[:, :, None, None]


class GConv(_Gated):
    """Gated convolution block (DSUNet.py:134-172)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1,
                 type_norm="BatchNorm2d", activation="ReLU", dropout=0.0):
        super().__init__()
        self.c = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                           groups=groups)
        self.m = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                         padding=padding, groups=groups), nn.Sigmoid())
        self.norm = _gconv_norm(type_norm, out_channels)
        self.activation = _gconv_activation(activation)
        self._twin = self._twin_key = None

    def forward(self, x, gate=None):
        return self._gated(x, None, self.norm, self.activation, gate)


class TrGConv(_Gated):
    """Gated transposed convolution (DSUNet.py:241-275); as in the reference, its norm and
    activation are constructed but not applied."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, groups=1,
                 type_norm="BatchNorm2d", activation="ReLU", dropout=0.0):
        super().__init__()
        self.c = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                    padding=padding, output_padding=output_padding, groups=groups)
        self.m = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                                  padding=padding, output_padding=output_padding, groups=groups),
                               nn.Sigmoid())
        self.norm_point = _gconv_norm(type_norm, out_channels)
        self.activation = _gconv_activation(activation)
        self._twin = self._twin_key = None

    def forward(self, x, skip=None):
        return self._gated(x, skip, None, None, None)


class TGRUBlock(nn.Module):
    """GRU along time for every (batch, frequency) row (one persistent recurrence launch, unet_amd.gru), then 1x1
    conv + BatchNorm + ReLU as one fused launch (DSUNet.py:281-312)."""

    def __init__(self, in_channels, hidden_size, out_channels, skipGRU=False, **kwargs):
        super().__init__()
        if skipGRU:
            raise Exception("Not Implemented")
        self.GRU = nn.GRU(in_channels, hidden_size, batch_first=True)
        self.conv = nn.Conv2d(hidden_size, out_channels, kernel_size=1)
        self.bn = nn.BatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.ReLU()
        self._fused = None

    def forward(self, x, rnn_state=None):
        require_device_tensor(x, "TGRUBlock")
        B, C, F, T = x.shape
        x_ = x.permute(0, 2, 3, 1).reshape(B * F, T, C)
        y_, rnn_state = _gru.run(self.GRU, x_.float(), rnn_state)
        y2 = y_.reshape(B, F, T, self.hidden_size).permute(0, 3, 1, 2).contiguous().to(x.dtype)
        if self.training:                    # the GRU above ran torch's device GRU (autograd); conv + BN + ReLU: one node
            from .train import conv_norm_act
            return conv_norm_act(self.conv, self.bn, _lib.ACT_RELU, None, y2)
        if self._fused is None:
            object.__setattr__(self, "_fused", FusedConv(self.conv, self.bn, self.relu))
        if not self._fused.fusable():
            raise NotImplementedError("TGRUBlock: BatchNorm2d only")
        return self._fused(y2)


class Attractor(nn.Module):
    """Steering vector + angle -> 256-dim code (DSUNet.py:11-43); plain torch ops."""

    def __init__(self, n_ch=4, n_fft=512):
        super().__init__()
        bins = n_fft // 2 + 1
        self.enc_SV = nn.Linear(n_ch * (n_fft + 2), bins)
        self.act_SV = nn.PReLU()
        self.enc_theta = nn.Linear(2, bins)
        self.act_theta = nn.Sigmoid()
        self.bn = nn.BatchNorm1d(bins)
        self.enc_2 = nn.Linear(bins, n_fft + 2)
        self.act_2 = nn.PReLU()
        self.enc_3 = nn.Linear(n_fft + 2, n_fft // 2)
        self.act_3 = nn.Sigmoid()

    def forward(self, SV, theta):
        h = self.act_SV(self.enc_SV(SV.reshape(SV.shape[0], -1))) + self.act_theta(self.enc_theta(theta))
        return self.act_3(self.enc_3(self.act_2(self.enc_2(self.bn(h)))))


class AttractEncoder(nn.Module):
    """code -> per-channel gate in [1/C, 1] (DSUNet.py:45-60)."""

    def __init__(self, n_ch, n_fft=512):
        super().__init__()
        self.n_ch = n_ch
        self.enc = nn.Linear(n_fft // 2, n_ch)
        self.conv = nn.Conv1d(1, n_ch, kernel_size=1)     # constructed, never used (as in the reference)
        self.act = nn.Sigmoid()

    def forward(self, attract):
        return torch.clamp(self.act(self.enc(attract)), min=1 / self.n_ch)


class _ComplexHead(nn.Module):
    """CRM / CRM_sig / CRM_soft / CM (DSUNet.py:351-402): squash, then channels (0, 1)
    become one complex mask; ``output`` applies it."""

    def __init__(self, squash, apply_mask=True):
        super().__init__()
        self.squash, self.apply_mask = squash, apply_mask
        if isinstance(squash, nn.Softplus):
            self.masking = squash                                # CRM_soft registers it under this name

    def forward(self, x):
        m = x if self.squash is None else self.squash(x)
        return torch.complex(m[:, 0, ...], m[:, 1, ...])

    def output(self, X, M):
        return X * M if self.apply_mask else M


def _mask_head(type_masking):
    if type_masking == "CRM":
        return _ComplexHead(nn.Tanh())
    if type_masking == "CRM_sig":
        return _ComplexHead(nn.Sigmoid())
    if type_masking == "CRM_soft":
        return _ComplexHead(nn.Softplus())
    if type_masking == "CM":
        return _ComplexHead(None, apply_mask=False)
    raise Exception("Not Implemented")


class DSUNet(nn.Module):
    def __init__(self, architecture, input_channels=8, n_fft=512, model_complexity=45, bottleneck="None",
                 padding_mode="zeros", type_masking="CRM", activation="LeakyReLU", dropout=0.0,
                 type_norm="BatcNorm2d", type_encoder="Encoder", type_residual="None"):
        super().__init__()
        if type_encoder == "Encoder":
            enc_cls, dec_cls = Encoder, Decoder
        elif type_encoder == "GConv":
            enc_cls, dec_cls = GConv, TrGConv
        else:
            raise Exception("Not Implemented Encoder")       # EncoderDWN: fails in the reference for in != out channels
        if bottleneck not in ("None", "TGRU") or type_residual != "None":
            raise Exception("Not Implemented")
        self.bottleneck = bottleneck
        self.n_channel = input_channels
        print("DSUNet::complexity {}".format(model_complexity))
        self.dropout = dropout
        L = self.n_layer = len(architecture["encoder"])
        enc_kw = [architecture["encoder"]["enc{}".format(i + 1)] for i in range(L)]
        encs = [enc_cls(**kw) for kw in enc_kw]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [dec_cls(**architecture["decoder"]["dec{}".format(L - i)]) for i in range(L)]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        self.BTN = TGRUBlock(128, 128, 128) if bottleneck == "TGRU" else nn.Identity()
        res = [nn.Identity() for _ in range(L - 1)]
        for i, m in enumerate(res):
            self.add_module("res{}".format(i), m)
        self.Attractor = Attractor(n_ch=4, n_fft=n_fft)
        self.attractEncoders = [AttractEncoder(n_ch=kw["out_channels"]) for kw in enc_kw]
        for i, m in enumerate(self.attractEncoders):
            self.add_module("attractEncoder_{}".format(i), m)
        self.add_module("linear", nn.Conv2d(architecture["decoder"]["dec1"]["out_channels"], 2, 1))
        self.padding_mode = padding_mode
        self.mask = _mask_head(type_masking)
        self.dr = nn.Dropout(self.dropout)
        self.decoders = nn.ModuleList(decs)
        self.encoders = nn.ModuleList(encs)
        self.attractors = nn.ModuleList(self.attractEncoders)
        self.res = nn.ModuleList(res)
        self._head = None
        self._blocked = None
        # bfloat16 inputs: run the convolutions with bfloat16 operands on the channel-blocked layout
        # (unet_amd/blocked.py) when every layer is one that kernel serves; False keeps the NCHW
        # bfloat16-storage kernels (fp32 products)
        self.bf16_matrix = True
        self.bf16_two_half_encoders = (0,)
        self.bf16_two_half_decoders = (L - 1,)
        # 16-bit format of the operands INSIDE that chain: "fp16" (default) keeps the network input and the first
        # encoder's operands bfloat16 (an STFT of unknown scale) and runs everything behind the first BatchNorm on fp16
        # operands -- same kernels, same matrix rate, three more mantissa bits; an activation beyond +-65504 becomes
        # infinity and shows as NaN in the mask (set "bf16" for such a model);
        # "bf16" keeps bfloat16 throughout (then the last decoder takes two weight halves as well).
        self.operand_format = "fp16"
        # a non-finite mask from the fp16 chain: unet_amd/blocked.py:fp16_overflow_guard ("defer" | "rerun" | "nan")
        self.on_fp16_overflow = "defer"

    def _blocked_layers(self):
        """BlockedConv per encoder / decoder / head, or False when a layer is outside that kernel."""
        fmt = self.operand_format
        if fmt not in ("fp16", "bf16"):
            raise ValueError("DSUNet.operand_format is 'fp16' or 'bf16'")
        e2 = tuple(self.bf16_two_half_encoders)
        d2 = tuple(self.bf16_two_half_decoders) if fmt == "bf16" else ()       # fp16 weights need no second half there
        key = (fmt, e2, d2)
        if self._blocked is None or self._blocked[0] != key:
            from .blocked import BlockedConv
            ok = isinstance(self.BTN, nn.Identity)
            for m in list(self.encoders) + list(self.decoders):
                conv = getattr(m, "conv", None) if isinstance(m, Encoder) else getattr(m, "transconv", None)
                code = activation_code(m.relu)        # the wide layers' epilogue has LeakyReLU / PReLU / ReLU only
                ok = ok and isinstance(m, (Encoder, Decoder)) and isinstance(m.bn, nn.BatchNorm2d) and \
                    code is not None and code[0] in (_lib.ACT_NONE, _lib.ACT_LEAKY, _lib.ACT_RELU) and \
                    conv.out_channels <= 128 and \
                    getattr(conv, "padding_mode", "zeros") == "zeros"
            built = False
            squash = activation_code(self.mask.squash)
            if ok and squash is not None and squash[0] != _lib.ACT_LEAKY:
                # Which layers take a weight as bfloat16 PLUS its rounding remainder (two MFMAs per product,
                # fp32-weight accuracy at twice the matrix work): by default the first encoder (bound by its
                # activation traffic, so the second MFMA is nearly free) and the last decoder (its error reaches the
                # mask without being averaged by later layers).  The head is not a layer of its own: the last decoder
                # keeps its result in the fp32 accumulators and applies the 1x1 head and the squash there
                # (BlockedConv.with_head), so neither that tensor nor the head's weights are ever rounded.
                # Measured at T=512, batch 32 (tools/dsunet_halves_probe.py full; max-norm error / ms), all bfloat16: no
                # layer 1.18e-2 / 6.07, these two 7.8e-3 / 6.82, the first two encoders and last two decoders
                # 7.0e-3 / 7.50, every layer 6.7e-3 / 10.2.  With fp16 operands behind the first encoder (whose own
                # weights keep their two bfloat16 halves: its input is bfloat16) no other layer needs a second half.
                built = dict(enc=[BlockedConv(m.conv, m.bn, m.relu, halves=2 if i in e2 else 1, out_fmt=fmt)
                                  for i, m in enumerate(self.encoders)],
                             dec=[BlockedConv(m.transconv, m.bn, m.relu, halves=2 if i in d2 else 1, out_fmt=fmt)
                                  for i, m in enumerate(self.decoders)],
                             head_act=squash[0])
            object.__setattr__(self, "_blocked", (key, built))
        return self._blocked[1]

    def _forward_blocked(self, layers, sf, attract):
        from .blocked import BlockedTensor
        L = self.n_layer
        x = BlockedTensor.from_nchw(sf.contiguous())
        skips = []
        for i, enc in enumerate(layers["enc"]):
            skips.append(x)
            x = enc(x, None, self.attractEncoders[i](attract).float().contiguous())
        for i, dec in enumerate(layers["dec"][:-1]):
            x = dec(x, None if i == 0 else skips[L - i])
        # last decoder + Conv2d(C, 2, 1) + squash in one launch, written as interleaved (re, im) pairs
        m = layers["dec"][-1].with_head(x, skips[1] if L > 1 else None, self.linear, layers["head_act"], interleaved=True)
        return torch.view_as_complex(m)                                # complex64 [B, F, T]

    def forward(self, sf, SV, theta):
        require_device_tensor(sf, "DSUNet")
        attract = self.Attractor(SV.float(), theta.float())            # tiny fp32 side network, whatever sf's storage
        L = self.n_layer
        if sf.dtype == torch.bfloat16 and self.bf16_matrix and not self.training:
            layers = self._blocked_layers()
            if layers:
                from .blocked import fp16_overflow_guard
                return fp16_overflow_guard(self, self._forward_blocked(layers, sf, attract),
                                           lambda: self._forward_blocked(self._blocked_layers(), sf, attract))
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(self.res[i](sf) if i != L - 1 else sf)
            gate = self.attractEncoders[i](attract).float().contiguous()      # [B, C_out]
            sf = enc(sf, gate=gate)                                    # dropout p=0 is the identity in eval
        p = self.BTN(sf)
        for i, dec in enumerate(self.decoders):
            p = dec(p) if i == 0 else dec(p, skips[L - i])
        if self.training:
            # differentiable head: the 1x1 convolution is an autograd node over HIP launches, the squash is torch's
            from .train import conv_norm_act
            m = conv_norm_act(self.linear, None, _lib.ACT_NONE, None, p)
            if self.mask.squash is not None:
                m = self.mask.squash(m)
            return torch.complex(m[:, 0, ...], m[:, 1, ...])
        if self._head is None:
            object.__setattr__(self, "_head", FusedConv(self.linear, None, self.mask.squash))
        m = self._head(p).float()                                      # [B, 2, F, T], squashed (complex64 output)
        return torch.complex(m[:, 0, ...], m[:, 1, ...])

    def masking(self, X, M):
        return self.mask.output(X, M)


class DSUNet_helper(nn.Module):
    """Wave in, wave out: STFT -> spectral / steering-vector / angle features -> DSUNet -> complex mask
    -> inverse STFT, with the reference's constructor and ``forward(x, angle, mic_pos)``
    (DSUNet.py:634-790).  The two transforms run on the convolution kernels (unet_amd/stft.py), the
    per-frequency Python loops of the reference's steering vector are closed-form tensor expressions,
    and -- as in the reference -- the source distance is one Frobenius norm over the whole batch
    (DSUNet.py:748), so all samples of a call share their delays.  Those twelve path lengths are computed on
    the host with the reference's operations (see ``_tdoa``); everything else runs on the device."""

    def __init__(self, architecture, n_fft=512, dropout=0.0, bottleneck="None", model_complexity=45, use_SV=True,
                 mag_phase=False, corr=False, DSB=False, activation="LeakyReLU", type_norm="BatchNorm2d",
                 type_masking="CRM", type_encoder="Encoder", type_residual="None"):
        super().__init__()
        from .stft import DeviceSTFT
        self.n_fft = n_fft
        self.n_hfft = n_fft // 2 + 1
        self.n_channel = 4
        model_channel = self.n_channel * 2 + (2 if DSB else 0)
        self.mag_phase, self.use_SV, self.corr, self.DSB = mag_phase, use_SV, corr, DSB
        self.net = DSUNet(architecture=architecture, input_channels=model_channel, dropout=dropout,
                          bottleneck=bottleneck, model_complexity=model_complexity, activation=activation,
                          type_norm=type_norm, type_masking=type_masking, type_encoder=type_encoder,
                          type_residual=type_residual)
        self.sr, self.ss, self.dist = 16000, 340.4, 100
        self.window = torch.hann_window(self.n_fft)
        self._stft = DeviceSTFT(n_fft, self.window)

    # True: the spectral feature is handed to the network as bfloat16, which selects its bfloat16-operand path
    # (STFT, features, mask application and inverse STFT stay fp32); expect ~3e-3 relative L2 error on the mask
    bf16_matrix = False

    def forward(self, x, angle, mic_pos):
        require_device_tensor(x, "DSUNet_helper")
        B, C, L = x.shape
        X = self._stft.stft(x.reshape(B * C, L).contiguous())
        T = X.shape[2]
        X = torch.nn.functional.pad(X, (0, 16 - T % 16)).reshape(B, C, self.n_hfft, -1)
        feat_src = X
        if self.corr:
            feat_src = torch.cat([X[:, :1], X[:, :1] * X[:, 1:]], dim=1)
        # angle / mic_pos may live on the host (no device wait for the path-length differences then)
        tdoa = self._tdoa(angle, mic_pos, x.device)     # once per call: the steering vector and the beamformer share it
        SV = self.steering_vector(angle, mic_pos, x.device, tdoa=tdoa)
        SV = torch.cat((SV.real, SV.imag), -1)
        theta = self.anlge_pre(angle.to(x.device, non_blocking=True))
        parts = [feat_src.real, feat_src.imag]
        if self.DSB:
            dsb = self.delay_n_sum(feat_src, angle, mic_pos, tdoa=tdoa)
            parts.append(torch.stack((dsb.real, dsb.imag), dim=1))
        spectral_feature = torch.cat(parts, dim=1).contiguous()
        if self.bf16_matrix:            # opt-in: the network's convolutions with bfloat16 operands (8x faster)
            spectral_feature = spectral_feature.to(torch.bfloat16)
        mask = self.net(spectral_feature, SV, theta)
        Y = self.net.masking(feat_src[:, 0], mask)
        return self._stft.istft(Y.contiguous(), L)

    def anlge_pre(self, angle):
        rad = (-angle) / 180 * torch.pi
        return torch.stack((torch.sin(rad), torch.cos(rad)), 1)

    def _tdoa(self, angle, mic_pos, device):
        """Path-length differences [B, C] on ``device``.  Computed on the HOST with the reference's own operations
        (DSUNet.py:741-750): they are differences of ~100 m distances, so one fp32 ulp of a distance moves the
        phase at the top bin by 1e-3 rad -- a device reduction that rounds the norm differently would show up at
        that level in the beamformer channel.  Twelve numbers per call: pass ``angle`` / ``mic_pos`` as host tensors
        and nothing waits for the device; device tensors cost one synchronising copy per call.  Nothing is kept
        between calls (the inputs are per-call data: a recycled allocation would alias any address-keyed cache);
        ``forward`` computes the differences once and hands them to both users."""
        ang = angle.detach().float().cpu()
        mp = mic_pos.detach().float().cpu()
        B = ang.shape[0]
        loc_src = torch.zeros(B, 3)
        loc_src[:, 0] = self.dist * torch.cos((-ang) / 180 * torch.pi) * torch.sin(torch.tensor(90 / 180 * torch.pi))
        loc_src[:, 1] = self.dist * torch.sin((-ang) / 180 * torch.pi) * torch.sin(torch.tensor(90 / 180 * torch.pi))
        loc_src[:, 2] = self.dist * torch.cos(torch.tensor((90) / 180 * torch.pi))
        tdoa = torch.zeros(B, self.n_channel)
        for i in range(self.n_channel):
            tdoa[:, i] = torch.norm(mp[:, i] - loc_src)     # one norm over the whole [B,3] difference, as in the reference
        return (tdoa[:, :] - tdoa[:, 0:1]).to(device)

    def _phasors(self, angle, mic_pos, sign, device=None, tdoa=None):
        device = angle.device if device is None else device
        if tdoa is None:
            tdoa = self._tdoa(angle, mic_pos, device)                                         # [B, C]
        const = sign * 2 * torch.pi * self.sr / (self.n_fft * self.ss)
        j = torch.arange(self.n_hfft, device=device, dtype=torch.float32)
        phase = tdoa[:, :, None] * j[None, None, :] * const                                   # [B, C, F]
        out = torch.polar(torch.ones_like(phase), phase)
        return out / torch.linalg.norm(out, dim=1, keepdim=True)

    def steering_vector(self, angle, mic_pos, device=None, tdoa=None):
        return self._phasors(angle, mic_pos, -1, device, tdoa)

    def delay_n_sum(self, X, angle, mic_pos, tdoa=None):
        return torch.einsum("bcft,bcf->bft", X, self._phasors(angle, mic_pos, +1, X.device, tdoa))
