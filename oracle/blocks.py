"""Building blocks of the oracle (CPU, stock torch ops).  Test infrastructure.

Every class states which reference block it restates.  Submodule names are
the reference's, so a reference ``state_dict`` loads unchanged.
"""

import torch
import torch.nn as nn


def _make_activation(name, where):
    """Activation by the reference's string switch (UNet_m.py:324-337)."""
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


def _norm_class(name):
    """Normalisation by the reference's string switch (UNet_m.py:312-319)."""
    table = {"BatchNorm2d": nn.BatchNorm2d, "InstanceNorm2d": nn.InstanceNorm2d,
             "LayerNorm": nn.LayerNorm}
    if name not in table:
        raise Exception("ERROR::Encoder: unknown normalization - {}".format(name))
    return table[name]


class Encoder(nn.Module):
    """Conv2d -> Dropout -> norm -> activation.  Restates UNet_m.py:304-345."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None,
                 dilation=1, norm="BatchNorm2d", padding_mode="zeros", dropout=0.0,
                 activation="LeakyReLU"):
        super().__init__()
        if padding is None:  # 'same' padding for odd kernels, UNet_m.py:307-308
            padding = [(k - 1) // 2 for k in kernel_size]
        norm_cls = _norm_class(norm)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size,
                              stride=stride, padding=padding, dilation=dilation,
                              padding_mode=padding_mode)
        self.bn = norm_cls(out_channels)
        self.acti = _make_activation(activation, "Encoder")
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.acti(self.bn(self.dropout(self.conv(x))))


class Decoder(nn.Module):
    """ConvTranspose2d -> norm -> activation.  Restates UNet_m.py:347-383."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=(0, 0),
                 dilation=1, output_padding=(0, 0), activation="LeakyReLU",
                 norm="BatchNorm2d"):
        super().__init__()
        norm_cls = _norm_class(norm)
        self.transconv = nn.ConvTranspose2d(in_channels, out_channels,
                                            kernel_size=kernel_size, stride=stride,
                                            dilation=dilation,
                                            output_padding=output_padding,
                                            padding=padding)
        self.bn = norm_cls(out_channels)
        self.acti = _make_activation(activation, "Encoder")

    def forward(self, x):
        return self.acti(self.bn(self.transconv(x)))


class ResBlock(nn.Module):
    """Five [depthwise 3x3 dilated along time -> pointwise 1x1 -> PReLU -> BN]
    layers; the first replaces x, the others add to the running value.
    Restates UNet_m.py:518-563.  ONE PReLU instance is shared by the five layers
    (UNet_m.py:528-529,551), so ``res_{0..4}.2.weight`` alias one tensor.
    """

    TIME_DILATIONS = (1, 2, 4, 8, 16)

    def __init__(self, n_channel, norm="BatchNorm2d", dropout=0.0, activation="PReLU"):
        super().__init__()
        if norm != "BatchNorm2d":
            raise Exception("ERROR::ResBlock: unknown nromalization - {}".format(norm))
        if activation != "PReLU":
            # the reference's GELU branch leaves its local unset and fails at use
            # (UNet_m.py:530-531,551); anything else raises at :533
            raise Exception("ERROR::ResBlock:Unknown activation type " + str(activation))
        shared_act = nn.PReLU()
        self.layers = []
        for i, d in enumerate(self.TIME_DILATIONS):
            layer = nn.Sequential(
                nn.Conv2d(n_channel, n_channel, (3, 3), (1, 1), padding=(1, d),
                          dilation=(1, d), groups=n_channel),
                nn.Conv2d(n_channel, n_channel, (1, 1)),
                shared_act,
                nn.BatchNorm2d(n_channel),
            )
            self.add_module("res_{}".format(i), layer)
            self.layers.append(layer)

    def forward(self, x):
        r = self.layers[0](x)
        for layer in self.layers[1:]:
            r = torch.add(r, layer(r))
        return r


# ----------------------------------------------------------------------------
# DSUNet's own block copies (different ctor signature from UNet_m's)
# ----------------------------------------------------------------------------

def _ds_activation(name):
    """DSUNet.py:79-86 / :187-194 -- no error branch: unknown names leave the
    attribute unset in the reference; here they raise KeyError at construction."""
    return {"LeakyReLU": lambda: nn.LeakyReLU(inplace=True), "PReLU": nn.PReLU,
            "SiLU": nn.SiLU, "Softplus": nn.Softplus}[name]()


def _ds_norm(type_norm):
    return {"BatchNorm2d": nn.BatchNorm2d, "InstanceNorm2d": nn.InstanceNorm2d}[type_norm]


class DSEncoder(nn.Module):
    """DSUNet's Encoder: conv -> dropout -> norm -> act.  Restates DSUNet.py:62-93."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=None,
                 padding_mode="zeros", activation="LeakyReLU", dropout=0.0,
                 type_norm="BatchNorm2d"):
        super().__init__()
        if padding is None:
            padding = [(k - 1) // 2 for k in kernel_size]
        self.dr = nn.Dropout(dropout)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size,
                              stride=stride, padding=padding, padding_mode=padding_mode)
        self.bn = _ds_norm(type_norm)(out_channels)
        self.relu = _ds_activation(activation)

    def forward(self, x):
        return self.relu(self.bn(self.dr(self.conv(x))))


class DSDecoder(nn.Module):
    """DSUNet's Decoder: tconv -> norm -> act; ``output_padding`` is positional
    BEFORE ``padding``.  Restates DSUNet.py:174-200."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, output_padding,
                 padding=(0, 0), activation="LeakyReLU", type_norm="BatchNorm2d"):
        super().__init__()
        self.transconv = nn.ConvTranspose2d(in_channels, out_channels,
                                            kernel_size=kernel_size, stride=stride,
                                            output_padding=output_padding,
                                            padding=padding)
        self.bn = _ds_norm(type_norm)(out_channels)
        self.relu = _ds_activation(activation)

    def forward(self, x):
        return self.relu(self.bn(self.transconv(x)))


class Attractor(nn.Module):
    """MLP from steering vector + angle to a 256-dim code.  Restates DSUNet.py:11-43."""

    def __init__(self, n_ch=4, n_fft=512):
        super().__init__()
        n_bins = n_fft // 2 + 1
        self.enc_SV = nn.Linear(n_ch * (n_fft + 2), n_bins)
        self.act_SV = nn.PReLU()
        self.enc_theta = nn.Linear(2, n_bins)
        self.act_theta = nn.Sigmoid()
        self.bn = nn.BatchNorm1d(n_bins)
        self.enc_2 = nn.Linear(n_bins, n_fft + 2)
        self.act_2 = nn.PReLU()
        self.enc_3 = nn.Linear(n_fft + 2, n_fft // 2)
        self.act_3 = nn.Sigmoid()

    def forward(self, SV, theta):
        sv = self.act_SV(self.enc_SV(SV.reshape(SV.shape[0], -1)))
        th = self.act_theta(self.enc_theta(theta))
        h = self.bn(sv + th)
        h = self.act_2(self.enc_2(h))
        return self.act_3(self.enc_3(h))


class AttractEncoder(nn.Module):
    """Per-layer channel gate max(sigmoid(Linear(code)), 1/C).  Restates
    DSUNet.py:45-60 (the Conv1d member is constructed but never used)."""

    def __init__(self, n_ch, n_fft=512):
        super().__init__()
        self.n_ch = n_ch
        self.enc = nn.Linear(n_fft // 2, n_ch)
        self.conv = nn.Conv1d(1, n_ch, kernel_size=1)
        self.act = nn.Sigmoid()

    def forward(self, attract):
        return torch.clamp(self.act(self.enc(attract)), min=1 / self.n_ch)


class CRM(nn.Module):
    """Complex ratio mask head: tanh, channels (0,1) -> complex.  Restates
    DSUNet.py:351-362."""

    def forward(self, x):
        m = torch.tanh(x)
        return m[:, 0, ...] + 1j * m[:, 1, ...]

    def output(self, X, M):
        return X * M


def _gconv_norm(type_norm, channels):
    # DSUNet.py:144-149: any other name is the identity
    if type_norm == "BatchNorm2d":
        return nn.BatchNorm2d(channels)
    if type_norm == "InstanceNorm2d":
        return nn.InstanceNorm2d(channels, track_running_stats=True)
    return nn.Identity()


def _gconv_activation(name):
    # DSUNet.py:151-162 (no else branch: an unknown name leaves the attribute unset in the reference)
    return {"ReLU": nn.ReLU, "SiLU": nn.SiLU, "ELU": nn.ELU, "Softplus": nn.Softplus, "PReLU": nn.PReLU,
            "LeakyReLU": nn.LeakyReLU}[name]()


class GConv(nn.Module):
    """Gated convolution: sigmoid(conv_m(x)) * conv_c(x) -> norm -> activation.  Restates DSUNet.py:134-172."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, groups=1,
                 type_norm="BatchNorm2d", activation="ReLU", dropout=0.0):
        super().__init__()
        self.c = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                           groups=groups)
        self.m = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                                         padding=padding, groups=groups), nn.Sigmoid())
        self.norm = _gconv_norm(type_norm, out_channels)
        self.activation = _gconv_activation(activation)

    def forward(self, x):
        return self.activation(self.norm(self.m(x) * self.c(x)))


class TrGConv(nn.Module):
    """Gated transposed convolution.  Restates DSUNet.py:241-275: the norm and the activation are
    constructed but NOT applied in forward."""

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

    def forward(self, x):
        return self.m(x) * self.c(x)


class TGRUBlock(nn.Module):
    """GRU along time for every (batch, frequency) row, then 1x1 conv -> BN -> ReLU.  Restates DSUNet.py:281-312."""

    def __init__(self, in_channels, hidden_size, out_channels, skipGRU=False, **kwargs):
        super().__init__()
        if skipGRU:
            raise Exception("Not Implemented")
        self.GRU = nn.GRU(in_channels, hidden_size, batch_first=True)
        self.conv = nn.Conv2d(hidden_size, out_channels, kernel_size=1)
        self.bn = nn.BatchNorm2d(out_channels)
        self.hidden_size = hidden_size
        self.relu = nn.ReLU()

    def forward(self, x, rnn_state=None):
        B, C, F, T = x.shape
        x_ = x.permute(0, 2, 3, 1).reshape(B * F, T, C)
        y_, rnn_state = self.GRU(x_, rnn_state)
        y2 = y_.reshape(B, F, T, self.hidden_size).permute(0, 3, 1, 2)
        return self.relu(self.bn(self.conv(y2)))


class SEBlock(nn.Module):
    """Squeeze-and-excitation scale.  Restates DSUNet.py:317-346."""

    def __init__(self, in_channel, ratio=16):
        super().__init__()
        size_se = int(in_channel / ratio)
        self.pool = torch.nn.AdaptiveAvgPool2d((1, 1))
        self.squeeze = nn.Linear(in_channel, size_se)
        self.relu = nn.ReLU()
        self.excitation = nn.Linear(size_se, in_channel)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        x_ = self.pool(x).permute(0, 2, 3, 1)
        x_ = self.sigmoid(self.excitation(self.relu(self.squeeze(x_)))).permute(0, 3, 1, 2)
        return x * x_


class CRM_sig(nn.Module):
    """DSUNet.py:364-375."""

    def forward(self, x):
        m = torch.sigmoid(x)
        return m[:, 0, ...] + 1j * m[:, 1, ...]

    def output(self, X, M):
        return X * M


class CRM_soft(nn.Module):
    """DSUNet.py:377-390."""

    def __init__(self):
        super().__init__()
        self.masking = nn.Softplus()

    def forward(self, x):
        m = self.masking(x)
        return m[:, 0, ...] + 1j * m[:, 1, ...]

    def output(self, X, M):
        return X * M


class CM(nn.Module):
    """Complex mapping: the network output IS the spectrum.  DSUNet.py:393-402."""

    def forward(self, x):
        return x[:, 0, ...] + 1j * x[:, 1, ...]

    def output(self, X, M):
        return M


class DepthwiseSeparableConv2d(nn.Module):
    """Per-channel conv -> BN -> ReLU -> 1x1 conv -> BN -> ReLU.  Restates UNet_m.py:7-31."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   groups=in_channels)
        self.bn_depth = nn.BatchNorm2d(out_channels)
        self.pointwise = nn.Conv2d(out_channels, out_channels, kernel_size=1)
        self.bn_point = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        x = torch.relu(self.bn_depth(self.depthwise(x)))
        return torch.relu(self.bn_point(self.pointwise(x)))


class GlobalChannelLayerNorm(nn.Module):
    """One mean and one variance per sample over every other axis, then a per-channel gain and offset.
    Restates UNet_m.py:739-763."""

    def __init__(self, channel_size):
        super().__init__()
        self.channel_size = channel_size
        self.gamma = nn.Parameter(torch.ones(channel_size))
        self.beta = nn.Parameter(torch.zeros(channel_size))

    def forward(self, x):
        dims = list(range(1, x.dim()))
        mean = x.mean(dim=dims, keepdim=True)
        var = torch.pow(x - mean, 2).mean(dim=dims, keepdim=True)
        normed = (x - mean) / (var + 1e-8).sqrt()
        return (self.gamma * normed.transpose(1, -1) + self.beta).transpose(1, -1)


class MultiScaleConvBlock(nn.Module):
    """1x1 expand -> norm -> depthwise-separable 1x1 -> norm -> 1x1 reduce -> norm.  Restates UNet_m.py:771-788."""

    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.net = nn.Sequential(nn.Conv2d(c, 2 * c, kernel_size=1), GlobalChannelLayerNorm(2 * c),
                                 DepthwiseSeparableConv2d(2 * c, 2 * c, kernel_size=1), GlobalChannelLayerNorm(2 * c),
                                 nn.Conv2d(2 * c, c, kernel_size=1), GlobalChannelLayerNorm(c))

    def forward(self, x):
        return self.net(x)


def custom_atan2(y, x):
    """Four-quadrant arctangent from ``atan(y / (x + 1e-6))`` and sign tests.  Restates UNet_m.py:677-686."""
    pi = torch.tensor([3.141592653589793], dtype=torch.float64).to(y.device, y.dtype)
    ans = torch.atan(y / (x + 1e-6))
    ans += ((y > 0) & (x < 0)) * pi
    ans -= ((y < 0) & (x < 0)) * pi
    ans *= 1 - ((y > 0) & (x == 0)) * 1.0
    ans += ((y > 0) & (x == 0)) * (pi / 2)
    ans *= 1 - ((y < 0) & (x == 0)) * 1.0
    ans += ((y < 0) & (x == 0)) * (-pi / 2)
    return ans


class MEA(nn.Module):
    """Mask estimation and application: three (3,1) convolutions give a ``mag_f_dim``-row magnitude mask, a
    real and an imaginary mask; ``output`` gathers each bin's magnitude with its frequency neighbours,
    weights them by the rectified magnitude mask, then rotates and scales by the complex mask.
    Restates UNet_m.py:691-732.  ``forward`` is pinned bit-exact; ``output`` is PARITY UNPINNED: the reference's
    own raises NameError (its ``custom_atan2`` uses numpy, which UNet_m.py never imports), so there is nothing to
    compare with -- it restates the arithmetic as written."""

    def __init__(self, in_channels=4, mag_f_dim=3):
        super().__init__()
        self.mag_mask = nn.Conv2d(in_channels, mag_f_dim, kernel_size=(3, 1), padding=(1, 0))
        self.real_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.imag_mask = nn.Conv2d(in_channels, 1, kernel_size=(3, 1), padding=(1, 0))
        self.register_buffer("kernel", torch.eye(mag_f_dim).reshape(mag_f_dim, 1, mag_f_dim, 1))
        self.mag_f_dim = mag_f_dim

    def forward(self, x):
        return (self.mag_mask(x), self.real_mask(x).squeeze(1), self.imag_mask(x).squeeze(1))

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
