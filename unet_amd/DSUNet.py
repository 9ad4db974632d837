"""Device ``DSUNet`` (directional separating U-Net) with the reference's constructor
and ``forward(sf, SV, theta)`` (DSUNet.py:467-631).

Every encoder is one fused launch: conv, eval-BatchNorm, activation AND the
per-(batch, channel) gate ``max(sigmoid(Linear(code)), 1/C)`` that the reference
multiplies in afterwards (DSUNet.py:599-602).  Decoders read ``[p, skip]`` as two
kernel inputs.  The 1x1 ``linear`` head with the complex-ratio-mask tanh is one
more fused launch.  The attractor MLP (0.8 MFLOP per utterance) runs as ordinary
torch device ops.  Supported variants: the constructor defaults
(type_encoder="Encoder", bottleneck="None", type_residual="None",
type_masking in CRM / CRM_sig / CRM_soft / CM).
"""

import torch
import torch.nn as nn

from .fused import FusedConv, require_device_tensor


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
            raise NotImplementedError("{}: train-mode forward is not available in this build; call .eval()"
                                      .format(type(self).__name__))
        if self._fused is None:
            object.__setattr__(self, "_fused", FusedConv(conv, self.bn, self.relu))
        if self._fused.fusable():
            return self._fused(x, skip, gate=gate)
        y = self.relu(self.bn(self._fused(x, skip, epilogue=False)))
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
        if type_encoder != "Encoder":
            raise Exception("Not Implemented Encoder")       # EncoderDWN / GConv: not built here
        if bottleneck != "None" or type_residual != "None":
            raise Exception("Not Implemented")
        self.bottleneck = bottleneck
        self.n_channel = input_channels
        print("DSUNet::complexity {}".format(model_complexity))
        self.dropout = dropout
        L = self.n_layer = len(architecture["encoder"])
        enc_kw = [architecture["encoder"]["enc{}".format(i + 1)] for i in range(L)]
        encs = [Encoder(**kw) for kw in enc_kw]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [Decoder(**architecture["decoder"]["dec{}".format(L - i)]) for i in range(L)]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        self.BTN = nn.Identity()
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

    def forward(self, sf, SV, theta):
        require_device_tensor(sf, "DSUNet")
        attract = self.Attractor(SV.float(), theta.float())            # tiny fp32 side network, whatever sf's storage
        L = self.n_layer
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(self.res[i](sf) if i != L - 1 else sf)
            gate = self.attractEncoders[i](attract).float().contiguous()      # [B, C_out]
            sf = enc(sf, gate=gate)                                    # dropout p=0 is the identity in eval
        p = self.BTN(sf)
        for i, dec in enumerate(self.decoders):
            p = dec(p) if i == 0 else dec(p, skips[L - i])
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
    (DSUNet.py:748), so all samples of a call share their delays."""

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

    def forward(self, x, angle, mic_pos):
        require_device_tensor(x, "DSUNet_helper")
        B, C, L = x.shape
        X = self._stft.stft(x.reshape(B * C, L).contiguous())
        T = X.shape[2]
        X = torch.nn.functional.pad(X, (0, 16 - T % 16)).reshape(B, C, self.n_hfft, -1)
        feat_src = X
        if self.corr:
            feat_src = torch.cat([X[:, :1], X[:, :1] * X[:, 1:]], dim=1)
        SV = self.steering_vector(angle, mic_pos)
        SV = torch.cat((SV.real, SV.imag), -1)
        theta = self.anlge_pre(angle)
        parts = [feat_src.real, feat_src.imag]
        if self.DSB:
            dsb = self.delay_n_sum(feat_src, angle, mic_pos)
            parts.append(torch.stack((dsb.real, dsb.imag), dim=1))
        spectral_feature = torch.cat(parts, dim=1).contiguous()
        mask = self.net(spectral_feature, SV, theta)
        Y = self.net.masking(feat_src[:, 0], mask)
        return self._stft.istft(Y.contiguous(), L)

    def anlge_pre(self, angle):
        rad = (-angle) / 180 * torch.pi
        return torch.stack((torch.sin(rad), torch.cos(rad)), 1)

    def _phasors(self, angle, mic_pos, sign):
        rad = (-angle) / 180 * torch.pi
        loc_src = torch.stack((self.dist * torch.cos(rad), self.dist * torch.sin(rad), torch.zeros_like(rad)), dim=1)
        loc_src[:, 2] = self.dist * torch.cos(torch.tensor(torch.pi / 2))
        # one norm over the whole [B,3] difference per microphone, as the reference computes it
        dist = torch.stack([torch.linalg.norm(mic_pos[:, i] - loc_src) for i in range(self.n_channel)])
        tdoa = (dist - dist[0])[None, :].expand(angle.shape[0], -1)                          # [B, C]
        const = sign * 2 * torch.pi * self.sr / (self.n_fft * self.ss)
        j = torch.arange(self.n_hfft, device=angle.device, dtype=torch.float32)
        phase = tdoa[:, :, None] * j[None, None, :] * const                                   # [B, C, F]
        out = torch.polar(torch.ones_like(phase), phase)
        return out / torch.linalg.norm(out, dim=1, keepdim=True)

    def steering_vector(self, angle, mic_pos):
        return self._phasors(angle, mic_pos, -1)

    def delay_n_sum(self, X, angle, mic_pos):
        return torch.einsum("bcft,bcf->bft", X, self._phasors(angle, mic_pos, +1))
