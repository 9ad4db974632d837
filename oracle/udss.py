"""Complex-valued U-Net family of the oracle (CPU, stock torch ops).  Test infrastructure.

Restates the reference's ``UDSS`` network (UDSS.py:420-683) and the complex blocks it is made of
(UNet_m.py:88-132, :214-225; UDSS.py:39-70, :91-116, :349-418).  Tensors are ``[B, C, F, T, 2]`` with the
real / imaginary parts innermost.  Registered submodule names equal the reference's, so its
``state_dict`` loads with ``strict=True``.
"""

import torch
import torch.nn as nn

from .blocks import Attractor, AttractEncoder


class ComplexConv2d(nn.Module):
    """Two real convolutions combined as a complex product; each carries its own bias, so the real part
    ends up with b_re - b_im and the imaginary part with b_re + b_im.  Restates UNet_m.py:88-102."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 **kwargs):
        super().__init__()
        kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias, **kwargs)
        self.conv_re = nn.Conv2d(in_channel, out_channel, kernel_size, **kw)
        self.conv_im = nn.Conv2d(in_channel, out_channel, kernel_size, **kw)

    def forward(self, x):
        xr, xi = x[..., 0], x[..., 1]
        re = self.conv_re(xr) - self.conv_im(xi)
        im = self.conv_re(xi) + self.conv_im(xr)
        return torch.stack((re, im), dim=-1)


class ComplexConvTranspose2d(nn.Module):
    """Transposed counterpart.  Restates UNet_m.py:104-132."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, bias=True, **kwargs):
        super().__init__()
        kw = dict(kernel_size=kernel_size, stride=stride, padding=padding, output_padding=output_padding,
                  groups=groups, bias=bias, dilation=dilation, **kwargs)
        self.tconv_re = nn.ConvTranspose2d(in_channel, out_channel, **kw)
        self.tconv_im = nn.ConvTranspose2d(in_channel, out_channel, **kw)

    def forward(self, x):
        xr, xi = x[..., 0], x[..., 1]
        re = self.tconv_re(xr) - self.tconv_im(xi)
        im = self.tconv_re(xi) + self.tconv_im(xr)
        return torch.stack((re, im), dim=-1)


class ComplexBatchNorm2d(nn.Module):
    """Independent BatchNorm2d on the real and on the imaginary part.  Restates UNet_m.py:214-225."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, **kwargs):
        super().__init__()
        kw = dict(num_features=num_features, momentum=momentum, affine=affine, eps=eps,
                  track_running_stats=track_running_stats, **kwargs)
        self.bn_re = nn.BatchNorm2d(**kw)
        self.bn_im = nn.BatchNorm2d(**kw)

    def forward(self, x):
        return torch.stack((self.bn_re(x[..., 0]), self.bn_im(x[..., 1])), dim=-1)


class ComplexInstanceNorm2d(nn.Module):
    """Restates UNet_m.py:227-237."""

    def __init__(self, num_features):
        super().__init__()
        self.in_re = nn.InstanceNorm2d(num_features=num_features)
        self.in_im = nn.InstanceNorm2d(num_features=num_features)

    def forward(self, x):
        return torch.stack((self.in_re(x[..., 0]), self.in_im(x[..., 1])), dim=-1)


def _norm_cls(type_norm):
    # UDSS.py:358-361: any other name leaves the class unbound there (UnboundLocalError at construction)
    return {"BatchNorm2d": ComplexBatchNorm2d, "InstanceNorm2d": ComplexInstanceNorm2d}[type_norm]


def _activation(name):
    # UDSS.py:374-379
    return {"LeakyReLU": lambda: nn.LeakyReLU(inplace=True), "PReLU": nn.PReLU, "SiLU": nn.SiLU}[name]()


class UEncoder(nn.Module):
    """Complex conv -> dropout -> complex norm -> activation on both parts.  Restates UDSS.py:349-386
    (complex=True)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None, complex=True,
                 padding_mode="zeros", activation="LeakyReLU", dropout=0.0, type_norm="BatchNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is DSUNet")
        if padding is None:
            padding = [(k - 1) // 2 for k in kernel_size]
        self.dr = nn.Dropout(dropout)
        self.conv = ComplexConv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                  padding=padding, padding_mode=padding_mode)
        self.bn = _norm_cls(type_norm)(out_channels)
        self.relu = _activation(activation)

    def forward(self, x):
        return self.relu(self.bn(self.dr(self.conv(x))))


class UDecoder(nn.Module):
    """Complex transposed conv -> complex norm -> activation; ``output_padding`` precedes ``padding``.
    Restates UDSS.py:388-418 (complex=True)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, output_padding, padding=(0, 0), complex=True,
                 activation="LeakyReLU", type_norm="BatchNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is DSUNet")
        self.transconv = ComplexConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                                output_padding=output_padding, padding=padding)
        self.bn = _norm_cls(type_norm)(out_channels)
        self.relu = _activation(activation)

    def forward(self, x):
        return self.relu(self.bn(self.transconv(x)))


class UTGRUBlock(nn.Module):
    """GRU along time over the interleaved (channel, re/im) features of every (batch, frequency) row, then
    complex 1x1 conv + complex BatchNorm + ReLU.  Restates UDSS.py:39-70."""

    def __init__(self, in_channels, hidden_size, out_channels, skipGRU=False, **kwargs):
        super().__init__()
        if skipGRU:
            raise Exception("Not Implemented")
        self.GRU = nn.GRU(in_channels * 2, hidden_size * 2, batch_first=True)
        self.conv = ComplexConv2d(hidden_size, out_channels, kernel_size=1)
        self.bn = ComplexBatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.ReLU()

    def forward(self, x, rnn_state=None):
        B, C, F, T, _ = x.shape
        seq = x.permute(0, 2, 3, 1, 4).reshape(B * F, T, C * 2)
        y, rnn_state = self.GRU(seq, rnn_state)
        y = y.reshape(B, F, T, self.hidden_size, 2).permute(0, 3, 1, 2, 4)
        return self.relu(self.bn(self.conv(y)))


class UFGRUBlock(nn.Module):
    """Bidirectional GRU along FREQUENCY over the interleaved (channel, re/im) features of every (batch, frame),
    then complex 1x1 conv + complex BatchNorm + PReLU.  Restates UDSS.py:15-37."""

    def __init__(self, in_channels, hidden_size, out_channels):
        super().__init__()
        self.GRU = nn.GRU(in_channels * 2, hidden_size * 2, batch_first=True, bidirectional=True)
        self.conv = ComplexConv2d(hidden_size * 2, out_channels, kernel_size=1)
        self.bn = ComplexBatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.PReLU()

    def forward(self, x):
        B, C, F, T, _ = x.shape
        seq = x.permute(0, 3, 2, 1, 4).reshape(B * T, F, C * 2)
        y, _ = self.GRU(seq)
        y = y.reshape(B, T, F, self.hidden_size * 2, 2).permute(0, 3, 2, 1, 4)
        return self.relu(self.bn(self.conv(y)))


class UFTGRUBlock(nn.Module):
    """Frequency GRU block, then time GRU block.  Restates UDSS.py:72-83."""

    def __init__(self, in_channels, hidden_size, out_channels, **kwargs):
        super().__init__()
        self.FGRU = UFGRUBlock(in_channels, hidden_size, out_channels, **kwargs)
        self.TGRU = UTGRUBlock(in_channels, hidden_size, out_channels, **kwargs)

    def forward(self, x, rnn_state=None):
        return self.TGRU(self.FGRU(x), rnn_state)


class UcRNN(nn.Module):
    """Complex recurrent bottleneck from four real GRUs (rr, ri, ir, ii) over the flattened (channel, frequency)
    features, BatchNorm1d and PReLU per part.  Restates UDSS.py:190-239 (style "GRU"), including its last two
    lines: the permute (1, 3, 2, 0) of the 4-D [B, C*F, T, 2] tensor yields [C*F, 2, T, B], and that buffer is
    reshaped to [B, C, F, T, 2] -- the axes come out scrambled, and so they do here."""

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

    def forward(self, x):
        B, C, F, T, _ = x.shape
        x = torch.reshape(torch.permute(x, (4, 0, 3, 1, 2)), (2, B, T, C * F))
        rr, ri = self.rr(x[0])[0], self.ri(x[0])[0]
        ir, ii = self.ir(x[1])[0], self.ii(x[1])[0]
        re = self.re_activation(self.re_norm(torch.permute(rr - ii, (0, 2, 1))))
        im = self.im_activation(self.im_norm(torch.permute(ri + ir, (0, 2, 1))))
        y = torch.permute(torch.stack((re, im), dim=-1), (1, 3, 2, 0))
        return torch.reshape(y, (B, C, F, T, 2))


class UCRM(nn.Module):
    """tanh, drop the singleton channel, re/im -> complex.  Restates UDSS.py:91-103."""

    def forward(self, x):
        m = torch.squeeze(torch.tanh(x), 1)
        return m[..., 0] + 1j * m[..., 1]

    def output(self, X, M):
        return X * M


class UCM(nn.Module):
    """Complex mapping: the network output is the result.  Restates UDSS.py:106-116."""

    def forward(self, x):
        m = torch.squeeze(x, 1)
        return m[..., 0] + 1j * m[..., 1]

    def output(self, X, M):
        return M


class UCMEA(nn.Module):
    """Complex mask estimation and application: three (3,1) convolutions on |x|, Re x, Im x give a magnitude mask
    and a complex mask; ``output`` applies sigmoid(magnitude mask) to the feature's magnitude, then the complex
    mask as sigmoid(|m|) and a phase rotation, with the reference's own atan2 (UDSS.py:120-177)."""

    def __init__(self, in_channels=1, mag_f_dim=3):
        super().__init__()
        self.mag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.real_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.imag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.mag_f_dim = mag_f_dim

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

    def forward(self, x):
        mag_mask = self.mag_mask(torch.sqrt(x[..., 0] ** 2 + x[..., 1] ** 2)).squeeze(1)
        real_mask = self.real_mask(x[..., 0]).squeeze(1)
        imag_mask = self.imag_mask(x[..., 1]).squeeze(1)
        return (mag_mask, real_mask, imag_mask)

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
    """Complex-valued directional source separation U-Net: ten complex encoders gated per channel by the
    attractor code, ten complex decoders fed [decoder, skip] concatenated on the channel axis, complex 1x1
    head to one channel, CRM / CM output.  Restates UDSS.py:420-580 for ``complex=True``,
    ``type_encoder="Complex"`` and the bottlenecks "None" / "TGRU" / "FTGRU" / "cRNN".

    The reference's default ``type_norm="BatcNorm2d"`` (sic) leaves the norm class unbound and fails at
    construction; callers pass ``type_norm="BatchNorm2d"``.  Decoders are built without ``type_norm`` and so
    always use BatchNorm (UDSS.py:475-478)."""

    def __init__(self, input_channels=4, n_fft=512, complex=True, model_complexity=45, bottleneck="None",
                 padding_mode="zeros", type_encoder="Complex", type_masking="CRM", activation="LeakyReLU",
                 dropout=0.0, type_norm="BatcNorm2d"):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is DSUNet")
        self.bottleneck = bottleneck
        self.complex = complex
        self.n_channel = input_channels
        model_complexity = int(model_complexity // 1.414)
        print("UDSS::complexity {}".format(model_complexity))
        for k, v in udss_table(model_complexity, input_channels).items():
            setattr(self, k, v)
        L = self.model_length = 10
        self.dropout = dropout
        if type_encoder != "Complex":
            raise Exception("Not Implemented")
        encs = [UEncoder(self.enc_channels[i], self.enc_channels[i + 1], kernel_size=self.enc_kernel_sizes[i],
                         stride=self.enc_strides[i], padding=self.enc_paddings[i], padding_mode=padding_mode,
                         dropout=dropout, activation=activation, type_norm=type_norm) for i in range(L)]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [UDecoder(self.dec_channels[i] + self.enc_channels[L - i], self.dec_channels[i + 1],
                         kernel_size=self.dec_kernel_sizes[i], stride=self.dec_strides[i],
                         padding=self.dec_paddings[i], output_padding=self.dec_output_paddings[i],
                         activation=activation) for i in range(L)]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        if bottleneck == "cRNN":
            self.BTN = UcRNN(128 * 2)
        elif bottleneck == "TGRU":
            self.BTN = UTGRUBlock(128, 256, 128)
        elif bottleneck == "FTGRU":
            self.BTN = UFTGRUBlock(128, 256, 128)
        elif bottleneck in ("RNN_AS", "AATT"):
            raise NotImplementedError("bottleneck {} is not restated".format(bottleneck))
        else:
            self.BTN = nn.Identity()
        self.Attractor = Attractor(n_ch=4, n_fft=n_fft)
        self.attractEncoders = [AttractEncoder(n_ch=self.enc_channels[i + 1]) for i in range(L)]
        for i, m in enumerate(self.attractEncoders):
            self.add_module("attractEncoder_{}".format(i), m)
        self.add_module("linear", ComplexConv2d(self.dec_channels[-1], 1, 1))
        self.padding_mode = padding_mode
        if type_masking == "CRM":
            self.mask = UCRM()
        elif type_masking == "CMEA":
            self.mask = UCMEA()
        elif type_masking == "CM":
            self.mask = UCM()
        else:
            raise Exception("Not Implemented")
        self.dr = nn.Dropout(self.dropout)
        self.decoders = nn.ModuleList(decs)
        self.encoders = nn.ModuleList(encs)
        self.attractors = nn.ModuleList(self.attractEncoders)

    def forward(self, sf, SV, theta):
        attract = self.Attractor(SV, theta)
        L = self.model_length
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(sf)
            sf = self.dr(enc(sf))
            gate = self.attractEncoders[i](attract)
            sf = torch.reshape(gate, (*gate.shape, 1, 1, 1)) * sf
        p = self.BTN(sf)
        for i, dec in enumerate(self.decoders):
            p = dec(p)
            if i == L - 1:
                break
            p = torch.cat([p, skips[L - 1 - i]], dim=1)
        return self.mask(self.linear(p))

    def masking(self, X, M):
        return self.mask.output(X, M)


class DAU(nn.Module):
    """The directional attractor U-Net of DTU.py:227-378: the same ten gated complex encoders, ten complex
    decoders and complex 1x1 head as ``UDSS`` (same layer table, DTU.py:380-477), BatchNorm only, bottleneck
    "None" only, and the mask always ``tanh`` of the head (DTU.py:355-358).  ``type_masking`` only selects the
    ``masking`` member (Identity for "CRM", the ``MEA`` module for "MEA"), which ``forward`` never calls.
    The reference's forward also prints the shape of every encoder output and gate; that is not restated."""

    def __init__(self, input_channels=4, n_fft=512, complex=True, model_complexity=45, bottleneck="None",
                 padding_mode="zeros", type_encoder="Complex", type_masking="CRM", activation="LeakyReLU",
                 dropout=0.0):
        super().__init__()
        if not complex:
            raise NotImplementedError("the real-valued form of this network is DSUNet")
        self.bottleneck = bottleneck
        self.complex = complex
        self.n_channel = input_channels
        model_complexity = int(model_complexity // 1.414)
        print("DAU::complexity {}".format(model_complexity))
        for k, v in udss_table(model_complexity, input_channels).items():
            setattr(self, k, v)
        L = self.model_length = 10
        self.dropout = dropout
        if type_encoder != "Complex":
            raise Exception("Not Implemented")
        if activation == "PReLU":
            raise TypeError("PReLU.__init__() got an unexpected keyword argument 'inplace'")     # DTU.py:181
        encs = [UEncoder(self.enc_channels[i], self.enc_channels[i + 1], kernel_size=self.enc_kernel_sizes[i],
                         stride=self.enc_strides[i], padding=self.enc_paddings[i], padding_mode=padding_mode,
                         dropout=dropout, activation=activation, type_norm="BatchNorm2d") for i in range(L)]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [UDecoder(self.dec_channels[i] + self.enc_channels[L - i], self.dec_channels[i + 1],
                         kernel_size=self.dec_kernel_sizes[i], stride=self.dec_strides[i],
                         padding=self.dec_paddings[i], output_padding=self.dec_output_paddings[i],
                         activation=activation) for i in range(L)]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        if bottleneck != "None":
            raise Exception("Not Implemented")
        self.BTN = nn.Identity()
        self.Attractor = Attractor(n_ch=4, n_fft=n_fft)
        self.attractEncoders = [AttractEncoder(n_ch=self.enc_channels[i + 1]) for i in range(L)]
        for i, m in enumerate(self.attractEncoders):
            self.add_module("attractEncoder_{}".format(i), m)
        self.add_module("linear", ComplexConv2d(self.dec_channels[-1], 1, 1))
        self.padding_mode = padding_mode
        if type_masking == "CRM":
            self.masking = nn.Identity()
        elif type_masking == "MEA":
            from .blocks import MEA
            self.masking = MEA()
        else:
            raise Exception("Not Implemented")
        self.dr = nn.Dropout(self.dropout)
        self.decoders = nn.ModuleList(decs)
        self.encoders = nn.ModuleList(encs)
        self.attractors = nn.ModuleList(self.attractEncoders)

    def forward(self, sf, SV, theta):
        attract = self.Attractor(SV, theta)
        L = self.model_length
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(sf)
            sf = self.dr(enc(sf))
            gate = self.attractEncoders[i](attract)
            sf = torch.reshape(gate, (*gate.shape, 1, 1, 1)) * sf
        p = self.BTN(sf)
        for i, dec in enumerate(self.decoders):
            p = dec(p)
            if i == L - 1:
                break
            p = torch.cat([p, skips[L - 1 - i]], dim=1)
        mask = torch.squeeze(torch.tanh(self.linear(p)), 1)
        return mask[..., 0] + 1j * mask[..., 1]
