"""Network assemblies of the oracle (CPU, stock torch ops).  Test infrastructure.

Restates the four ``forward`` bodies on the hot path.  Registered submodule
names equal the reference's so its ``state_dict`` loads unchanged.
"""

import torch
import torch.nn as nn

from .blocks import (Encoder, Decoder, ResBlock, DSEncoder, DSDecoder, Attractor,
                     AttractEncoder, CRM, CRM_sig, CRM_soft, CM, GConv, TrGConv, TGRUBlock, SEBlock,
                     MultiScaleConvBlock, MEA)
from .tables import unet20_layers


def _register_concat_shell(net, encoders, decoders, bottlenecks, res_paths):
    """Module registration shared by UNet and UNet20 (UNet.py:26-66,
    cRFUNet.py:227-267): ``encoder_i``/``decoder_i``, identity ``res_path{i}``
    (L entries: L-1 real + one dummy) and identity ``bottleneck0`` by default."""
    n = len(encoders)
    net.len_model = n
    net.encoders, net.decoders = encoders, decoders
    for i, m in enumerate(encoders):
        net.add_module("encoder_{}".format(i), m)
    for i, m in enumerate(decoders):
        net.add_module("decoder_{}".format(i), m)
    net.res_paths = []
    if res_paths is not None:
        if len(res_paths) != n - 1:
            raise Exception("ERROR::unmatched res_path : {} != {}".format(len(res_paths), n - 1))
        paths = list(res_paths)
    else:
        paths = [nn.Identity() for _ in range(n - 1)]
    paths.append(nn.Identity())  # slot for the last encoder's input; value never used
    for i, m in enumerate(paths):
        net.add_module("res_path{}".format(i), m)
        net.res_paths.append(m)
    net.bottlenecks = list(bottlenecks) if bottlenecks is not None else [nn.Identity()]
    for i, m in enumerate(net.bottlenecks):
        net.add_module("bottleneck{}".format(i), m)


def _concat_shell_forward(net, x):
    """Encoder loop, bottleneck, decoder loop with [decoder, skip] concat
    (UNet.py:85-106 == cRFUNet.py:290-311)."""
    skips = []
    for i, enc in enumerate(net.encoders):
        skips.append(net.res_paths[i](x))
        x = enc(x)
    p = x
    for b in net.bottlenecks:
        p = b(p)
    last = net.len_model - 1
    for i, dec in enumerate(net.decoders):
        p = dec(p)
        if i == last:
            break
        p = torch.cat([p, skips[last - i]], dim=1)
    return p


class UNet(nn.Module):
    """Generic concat-skip U-Net around caller-built block lists; 1x1 conv(1->1)
    + Sigmoid mask head.  Restates UNet.py:6-113.

    The reference constructor reads an undefined global ``print_shape``
    (UNet.py:24) and only works after that name is injected into its module;
    here it is a keyword that defaults to False.
    """

    def __init__(self, encoders, decoders, bottlenecks=None, res_paths=None,
                 device="cuda:0", print_shape=False):
        super().__init__()
        self.device = device
        if len(encoders) != len(decoders):
            raise Exception("ERROR::unmatched legnth : enc {} != dec {}".format(
                len(encoders), len(decoders)))
        self.print_shape = print_shape
        _register_concat_shell(self, encoders, decoders, bottlenecks, res_paths)
        self.add_module("linear", nn.Conv2d(1, 1, 1))
        self.activation_mask = nn.Sigmoid()

    def forward(self, x):
        p = _concat_shell_forward(self, x)
        return self.activation_mask(self.linear(p))


class UNet20(nn.Module):
    """cRFUNet's 10+10-layer net: fixed table, 1x1 head to 2*c_out*N channels,
    Softplus (or Sigmoid), then a RAW reshape of the NCHW buffer to
    [B,N,c_out,2L_f+1,2L_t+1,F,T,2] viewed as complex (adjacent TIME frames pair
    up as re/im).  Restates cRFUNet.py:162-324."""

    def __init__(self, c_in=4 + 6 + 4, c_out=4, L_t=0, L_f=0, n_target=4, n_feature=12,
                 n_fft=512, device="cuda:0", print_shape=False, mask="Softplus"):
        super().__init__()
        self.print_shape = print_shape
        self.N, self.C, self.c_out = n_target, c_in, c_out
        self.L_t, self.L_f = L_t, L_f
        self.F = int(n_fft / 2 + 1)
        self.T = 125  # hard-coded in the reference (cRFUNet.py:187); callers overwrite it
        f_dim = 32
        enc_rows, dec_rows = unet20_layers(c_in=c_in, c_last=2 * f_dim, f_dim=f_dim)
        encoders = [Encoder(*r["args"], **r["kw"]) for r in enc_rows]
        decoders = [Decoder(*r["args"], **r["kw"]) for r in dec_rows]
        self.linear = nn.Conv2d(2 * f_dim, 2 * c_out * n_target, 1)
        _register_concat_shell(self, encoders, decoders, None, None)
        self.activation_mask = nn.Sigmoid() if mask == "Sigmoid" else nn.Softplus()

    def forward(self, x):
        p = _concat_shell_forward(self, x)
        f = self.activation_mask(self.linear(p))
        f = torch.reshape(f, (f.shape[0], self.N, self.c_out, 2 * self.L_f + 1,
                              2 * self.L_t + 1, self.F, self.T, 2))
        return torch.view_as_complex(f)


class ResUNet(nn.Module):
    """Frequency-strided residual U-Net with additive skips through 1x1
    res-paths and a projected-LSTM bottleneck; output = (tanh(y)+1e-13) * input[:, :2].
    Restates ResUNet.py:22-220."""

    def __init__(self, c_in=4, c_out=1, n_fft=512, device="cuda:0", output="mapping",
                 print_shape=False, T=125):
        super().__init__()
        n_hfft = int(n_fft / 2 + 1)
        self.print_shape = print_shape
        self.F, self.T = n_hfft, T
        C = 64
        prelu = dict(activation="PReLU")

        # input layers act on [B,C,T,F]: conv along F, LayerNorm over F (ResUNet.py:47-54)
        self.layer_data = nn.Sequential(Encoder(c_in - 2, 20, (1, 3), 1, (0, 1), 1),
                                        nn.LayerNorm(n_hfft))
        self.layer_AF = nn.Sequential(Encoder(2, C - 20, (1, 3), 1, (0, 1), 1),
                                      nn.LayerNorm(n_hfft))

        def down():  # ResUNet.py:59-78
            return nn.Sequential(Encoder(C, C, (3, 3), (2, 1), (0, 1), **prelu), ResBlock(C))

        def up(pad, opad):  # ResUNet.py:94-113
            return nn.Sequential(ResBlock(C), Decoder(C, C, (4, 1), (2, 1), pad,
                                                      output_padding=opad, **prelu))

        self.encoders = [ResBlock(C)] + [down() for _ in range(5)] + [
            Encoder(C, 2 * C, (3, 1), (2, 1), (0, 0), **prelu),
            Encoder(2 * C, 6 * C, (3, 1), (1, 1), (0, 0), **prelu)]
        for i, m in enumerate(self.encoders):
            self.add_module("enc_{}".format(i), m)

        self.decoders = [
            Decoder(6 * C, 2 * C, (3, 1), (1, 1), (0, 0), **prelu),
            Decoder(2 * C, C, (4, 1), (2, 1), (1, 0), output_padding=(1, 0), **prelu),
            up((1, 0), (1, 0)), up((1, 0), (1, 0)), up((1, 0), (1, 0)),
            up((0, 0), (0, 0)), up((1, 0), (1, 0)),
            ResBlock(C)]
        for i, m in enumerate(self.decoders):
            self.add_module("dec_{}".format(i), m)
        self.len_model = len(self.encoders)

        widths = [C] * 6 + [2 * C, 6 * C]  # ResUNet.py:125-132
        self.res_paths = [Encoder(w, w, 1, 1, 0, 1, **prelu) for w in widths]
        for i, m in enumerate(self.res_paths):
            self.add_module("res_path_{}".format(i), m)

        self.bottleneck = nn.LSTM(6 * C, 10 * C, 3, batch_first=True, proj_size=6 * C)
        self.out_layer = nn.ConvTranspose2d(C, 2, (3, 1), stride=1, padding=(1, 0),
                                            dilation=1, output_padding=(0, 0))
        self.last_activation = nn.Tanh()

    def forward(self, input):
        feat = torch.permute(input[:, :-2, :, :], (0, 1, 3, 2))
        af = torch.permute(input[:, -2:, :, :], (0, 1, 3, 2))
        feat = torch.permute(self.layer_data(feat), (0, 1, 3, 2))
        af = torch.permute(self.layer_AF(af), (0, 1, 3, 2))
        x = torch.cat((feat, af), dim=1)

        res = []
        for enc in self.encoders:
            x = enc(x)
            res.append(x)

        # [B,C,1,T] -> [B,T,C] -> LSTM -> back (ResUNet.py:186-197)
        x = torch.reshape(x, (x.shape[0], x.shape[1], x.shape[3]))
        x = self.bottleneck(torch.permute(x, (0, 2, 1)))[0]
        x = torch.permute(x, (0, 2, 1))
        x = torch.reshape(x, (x.shape[0], x.shape[1], 1, x.shape[2]))

        for i, rp in enumerate(self.res_paths):
            res[i] = rp(res[i])

        y = x
        for i, dec in enumerate(self.decoders):
            y = dec(torch.add(y, res[-1 - i]))

        mask = self.last_activation(self.out_layer(y)) + 1e-13
        return torch.mul(mask, input[:, :2, :, :])


class ResUNetOnFreq(nn.Module):
    """30-channel variant whose strided convolutions are (3,1): frequency is halved per block, the
    time axis is only touched by the ResBlocks' dilated depthwise taps and the LSTM.  One input
    branch, ``n_block`` down blocks, LSTM(192, 300, 3, proj 192), a one-channel transposed output
    layer and a Softplus / Sigmoid mask that is returned as it is.  Restates ResUNet.py:227-389.

    With the shipped kernel table the frequency sizes only line up for n_block = 5 at 257 bins."""

    def __init__(self, c_in=4, c_out=1, n_fft=512, device="cuda:0", print_shape=False, T=125,
                 n_block=5, activation="Softplus", Softplus_thr=20, norm="BatchNorm2d", dropout=0.0):
        super().__init__()
        n_hfft = int(n_fft / 2 + 1)
        self.print_shape = print_shape
        self.F, self.T = n_hfft, T
        W = 30
        if n_block < 2:
            raise Exception("ERROR::ResUnetOnFreq : n_block({}) < 2".format(n_block))
        kw = dict(activation="PReLU", norm=norm)

        self.layer_input = nn.Sequential(Encoder(c_in, W, (1, 3), 1, (0, 1), 1), nn.LayerNorm(n_hfft))

        self.encoders = [ResBlock(W)]
        for _ in range(n_block):  # ResUNet.py:267-271
            self.encoders.append(nn.Sequential(Encoder(W, W, (3, 1), (2, 1), (0, 0), **kw), ResBlock(30)))
        self.encoders.append(Encoder(W, 64, (3, 1), (2, 1), (0, 0), **kw))
        self.encoders.append(Encoder(64, 192, (3, 1), (1, 1), (0, 0), **kw))
        for i, m in enumerate(self.encoders):
            self.add_module("enc_{}".format(i), m)

        def up(kh):  # ResUNet.py:288-300
            return nn.Sequential(ResBlock(W), Decoder(W, W, (kh, 1), (2, 1), (1, 0),
                                                      output_padding=(1, 0), **kw))

        self.decoders = [Decoder(192, 64, (3, 1), (1, 1), (0, 0), **kw),
                         Decoder(64, W, (4, 1), (2, 1), (1, 0), output_padding=(1, 0), **kw)]
        self.decoders += [up(4) for _ in range(n_block - 2)] + [up(5), up(4), ResBlock(W)]
        for i, m in enumerate(self.decoders):
            self.add_module("dec_{}".format(i), m)
        self.len_model = len(self.encoders)

        widths = [W] * (n_block + 1) + [64, 192]  # ResUNet.py:310-315
        self.res_paths = [Encoder(w, w, 1, 1, 0, 1, activation="PReLU") for w in widths]
        for i, m in enumerate(self.res_paths):
            self.add_module("res_path_{}".format(i), m)

        self.bottleneck = nn.LSTM(192, 300, 3, batch_first=True, proj_size=192, dropout=dropout)
        self.out_layer = nn.ConvTranspose2d(W, c_out, (3, 1), stride=1, padding=(1, 0),
                                            dilation=1, output_padding=(0, 0))
        if activation == "Softplus":
            self.activation_mask = nn.Softplus(threshold=Softplus_thr)
        elif activation == "Sigmoid":
            self.activation_mask = nn.Sigmoid()
        else:
            self.activation_mask = nn.Softplus()

    def forward(self, input):
        feat = self.layer_input(torch.permute(input, (0, 1, 3, 2)))
        x = torch.permute(feat, (0, 1, 3, 2))
        res = []
        for enc in self.encoders:
            x = enc(x)
            res.append(x)
        x = torch.reshape(x, (x.shape[0], x.shape[1], x.shape[3]))
        x = self.bottleneck(torch.permute(x, (0, 2, 1)))[0]
        x = torch.permute(x, (0, 2, 1))
        y = torch.reshape(x, (x.shape[0], x.shape[1], 1, x.shape[2]))
        for i, rp in enumerate(self.res_paths):
            res[i] = rp(res[i])
        for i, dec in enumerate(self.decoders):
            y = dec(torch.add(y, res[-1 - i]))
        return self.activation_mask(self.out_layer(y))

    def output(self, mask, feature):  # ResUNet.py:388-389
        return mask * feature[:, :2]


class LSTMBlock(nn.Module):
    """[B,C,F',T] -> [B,T,C*F'] -> projected LSTM -> back.  Restates UNet_m.py:587-610."""

    def __init__(self, n_dim, n_hidden, n_layer=3, proj_size=None, dropout=0.2):
        super().__init__()
        if proj_size is None:
            proj_size = n_dim
        self.rnn = nn.LSTM(n_dim, n_hidden, n_layer, batch_first=True, proj_size=proj_size, dropout=dropout)

    def forward(self, x):
        shape = x.shape
        seq = torch.permute(torch.reshape(x, (shape[0], shape[1] * shape[2], shape[3])), (0, 2, 1))
        y = torch.permute(self.rnn(seq)[0], (0, 2, 1))
        return torch.reshape(y, tuple(shape))


class ResUNetOnFreq2(nn.Module):
    """ResUNetOnFreq cut after its 30-channel blocks: six encoders down to 7 frequency rows, an LSTM over the
    flattened (channel, frequency) features, six decoders, one-channel mask.  Restates ResUNet.py:394-591 with
    the LSTM bottleneck.

    ``multi_scale=True`` registers average pools, a ``MultiScaleConvBlock`` and upsamplers and runs the pools
    and the block beside the encoders (ResUNet.py:447-467, :553-560); nothing downstream reads their result,
    so the network's output does not depend on them.  ``activation="MEA"`` puts the MEA head on the output
    (three masks); the reference's ``output`` then names an attribute that does not exist (ResUNet.py:589).

    The reference indexes the bottleneck's output tensor with ``[0]`` (ResUNet.py:565, a leftover of
    ``nn.LSTM``'s tuple): the batch axis is dropped, so the FIRST sample's bottleneck output is broadcast to
    every sample of the batch by the following add.  That behaviour is restated as it is."""

    def __init__(self, c_in=1, c_out=1, n_fft=512, device="cuda:0", print_shape=False, n_block=5,
                 activation="Softplus", bottleneck="LSTM", Softplus_thr=20, norm="BatchNorm2d", dropout=0.0,
                 activation_layer="PReLU", multi_scale=False):
        super().__init__()
        n_hfft = int(n_fft / 2 + 1)
        self.print_shape = print_shape
        self.activation = activation
        self.multi_scale = multi_scale
        self.F = n_hfft
        W = 30
        if n_block < 2:
            raise Exception("ERROR::ResUnetOnFreq : n_block({}) < 2".format(n_block))
        if bottleneck == "FTGRU":
            raise NotImplementedError("bottleneck FTGRU is not restated")
        kw = dict(activation=activation_layer, norm=norm)
        self.layer_input = nn.Sequential(Encoder(c_in, W, (1, 3), 1, (0, 1), 1), nn.LayerNorm(n_hfft))
        self.encoders = [ResBlock(W)] + [nn.Sequential(Encoder(W, W, (3, 1), (2, 1), (0, 0), **kw), ResBlock(30))
                                         for _ in range(n_block)]
        for i, m in enumerate(self.encoders):
            self.add_module("enc_{}".format(i), m)
        if multi_scale:
            self.ms = [nn.Sequential(nn.AvgPool2d((2 ** (n_block + 1 - i), 1), stride=(2 ** (n_block - i), 1),
                                                  padding=(1, 0))) for i in range(n_block)]
            for i, m in enumerate(self.ms):
                self.add_module("ms_{}".format(i), m)
            self.ms_block = MultiScaleConvBlock(W)
            self.upsample = [nn.Sequential(nn.Upsample(scale_factor=(s, 1), mode="nearest"), nn.Sigmoid())
                             for s in [1, 2.2, 4.5, 9, 18.4, 36.8]]
            for i, m in enumerate(self.upsample):
                self.add_module("up_{}".format(i), m)

        def up(kh):
            return nn.Sequential(ResBlock(W), Decoder(W, W, (kh, 1), (2, 1), (1, 0), output_padding=(1, 0), **kw))

        self.decoders = [up(4) for _ in range(n_block - 2)] + [up(5), up(4), ResBlock(W)]
        for i, m in enumerate(self.decoders):
            self.add_module("dec_{}".format(i), m)
        self.len_model = len(self.encoders)
        self.res_paths = [Encoder(W, W, 1, 1, 0, 1, activation=activation_layer) for _ in range(n_block + 1)]
        for i, m in enumerate(self.res_paths):
            self.add_module("res_path_{}".format(i), m)
        self.bottleneck = LSTMBlock(210, 300, n_layer=3, dropout=dropout)
        self.out_layer = nn.ConvTranspose2d(W, c_out, (3, 1), stride=1, padding=(1, 0), dilation=1,
                                            output_padding=(0, 0))
        if activation == "Softplus":
            self.activation_mask = nn.Softplus(threshold=Softplus_thr)
        elif activation == "Sigmoid":
            self.activation_mask = nn.Sigmoid()
        elif activation == "Tanh":
            self.activation_mask = nn.Tanh()
        elif activation == "Identity":
            self.activation_mask = nn.Identity()
        elif activation == "MEA":
            if c_in == 1:
                raise Exception("ERROR::ResUnetOnFreq::feature must be complex")
            self.activation_mask = MEA(in_channels=c_out)
            self.add_module("MEA", self.activation_mask)
        else:
            self.activation_mask = nn.Softplus()

    def forward(self, input):
        feat = self.layer_input(torch.permute(input, (0, 1, 3, 2)))
        x = torch.permute(feat, (0, 1, 3, 2))
        res = []
        ms = None
        for i, enc in enumerate(self.encoders):
            x = enc(x)
            res.append(x)
            if self.multi_scale and i < len(self.ms):
                if ms is None:
                    ms = self.ms[i](x)
                else:
                    ms += self.ms[i](x)
                ms = self.ms_block(ms)
        y = self.bottleneck(x)[0]                    # [C, F', T]: sample 0 only (see the class docstring)
        for i, rp in enumerate(self.res_paths):
            res[i] = rp(res[i])
        for i, dec in enumerate(self.decoders):
            y = dec(torch.add(y, res[-1 - i]))
        return self.activation_mask(self.out_layer(y))

    def output(self, mask, feature):
        if self.activation == "MEA":
            return self.last_activation.output(mask, feature)    # no such attribute: AttributeError, as there
        return mask * feature[:, :2]


class DSUNet(nn.Module):
    """Directional-separating U-Net built from an external architecture dict
    (keys ``encoder.enc{1..L}`` / ``decoder.dec{L..1}``); each encoder output is
    gated per (batch, channel) by a code of steering vector + angle; CRM head.
    Restates DSUNet.py:467-631 for type_encoder "Encoder" / "GConv", bottleneck "None" / "TGRU",
    type_residual "None" / "SE" (which the reference builds on the wrong channel count, so its forward
    raises), type_masking "CRM" / "CRM_sig" / "CRM_soft" / "CM"."""

    def __init__(self, architecture, input_channels=8, n_fft=512, model_complexity=45,
                 bottleneck="None", padding_mode="zeros", type_masking="CRM",
                 activation="LeakyReLU", dropout=0.0, type_norm="BatcNorm2d",
                 type_encoder="Encoder", type_residual="None"):
        super().__init__()
        if type_encoder == "Encoder":
            enc_cls, dec_cls = DSEncoder, DSDecoder
        elif type_encoder == "GConv":
            enc_cls, dec_cls = GConv, TrGConv
        else:
            raise Exception("Not Implemented Encoder")
        heads = {"CRM": CRM, "CRM_sig": CRM_sig, "CRM_soft": CRM_soft, "CM": CM}
        if type_masking not in heads:
            raise Exception("Not Implemented")
        self.bottleneck = bottleneck
        self.n_channel = input_channels
        self.dropout = dropout
        L = self.n_layer = len(architecture["encoder"])
        enc_kw = [architecture["encoder"]["enc{}".format(i + 1)] for i in range(L)]
        dec_kw = [architecture["decoder"]["dec{}".format(L - i)] for i in range(L)]

        encs = [enc_cls(**kw) for kw in enc_kw]
        for i, m in enumerate(encs):
            self.add_module("encoder{}".format(i), m)
        decs = [dec_cls(**kw) for kw in dec_kw]
        for i, m in enumerate(decs):
            self.add_module("decoder{}".format(i), m)
        self.BTN = TGRUBlock(128, 128, 128) if bottleneck == "TGRU" else nn.Identity()
        res = [SEBlock(enc_kw[i]["out_channels"]) if type_residual == "SE" else nn.Identity() for i in range(L - 1)]
        for i, m in enumerate(res):
            self.add_module("res{}".format(i), m)
        self.Attractor = Attractor(n_ch=4, n_fft=n_fft)
        self.attractEncoders = [AttractEncoder(n_ch=kw["out_channels"]) for kw in enc_kw]
        for i, m in enumerate(self.attractEncoders):
            self.add_module("attractEncoder_{}".format(i), m)
        self.add_module("linear", nn.Conv2d(architecture["decoder"]["dec1"]["out_channels"], 2, 1))
        self.padding_mode = padding_mode
        self.mask = heads[type_masking]()
        self.dr = nn.Dropout(self.dropout)
        # the reference re-registers the same modules in ModuleLists
        # (DSUNet.py:577-580), which doubles their state_dict keys
        self.decoders = nn.ModuleList(decs)
        self.encoders = nn.ModuleList(encs)
        self.attractors = nn.ModuleList(self.attractEncoders)
        self.res = nn.ModuleList(res)

    def forward(self, sf, SV, theta):
        attract = self.Attractor(SV, theta)
        L = self.n_layer
        skips = []
        for i, enc in enumerate(self.encoders):
            skips.append(self.res[i](sf) if i != L - 1 else sf)
            sf = self.dr(enc(sf))
            gate = self.attractEncoders[i](attract)
            sf = torch.reshape(gate, (*gate.shape, 1, 1)) * sf
        p = self.BTN(sf)
        for i, dec in enumerate(self.decoders):
            p = dec(p)
            if i == L - 1:
                break
            p = torch.cat([p, skips[L - 1 - i]], dim=1)
        return self.mask(self.linear(p))

    def masking(self, X, M):
        return self.mask.output(X, M)
