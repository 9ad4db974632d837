"""Device ``ResUNet`` with the reference's constructor and ``forward`` (ResUNet.py:22-220).

What runs where:
  * input layers: the (1,3) conv over the permuted [B,C,T,F] view is run as a (3,1)
    conv on the tensor as it is (weight axes swapped), BatchNorm + LeakyReLU in its
    epilogue; LayerNorm(257) over frequency is one launch per branch that writes
    straight into its channel range of the 64-channel tensor (no permute, no cat);
  * encoders / decoders / res-paths: fused conv launches (PReLU + BatchNorm in the
    epilogue, ``y + res`` read as two inputs), ResBlocks as in ``UNet_m.ResBlock``;
  * bottleneck: ``nn.LSTM(384, 640, 3, proj_size=384)``: the 501-step recurrence of each layer is one
    persistent launch (unet_amd/lstm.py), its time-parallel products are GEMMs;
  * output: transposed (3,1) conv to 2 channels, then (tanh(y) + 1e-13) * input[:, :2]
    as one launch.
"""

import torch
import torch.nn as nn

from .UNet_m import Decoder, Encoder, MEA, MultiScaleConvBlock, ResBlock
from .fused import FusedConv, require_device_tensor
from .lstm import lstm_forward, lstm_forward_train, trainable_on_device, supported as lstm_supported
from .ops import layernorm_f, tanh_mask_mul


def _trunk(net, x, bottleneck=None):
    """Encoders, projected-LSTM bottleneck, 1x1 res-paths and the additive-skip decoders shared by
    ResUNet, ResUNetOnFreq and ResUNetOnFreq2 (ResUNet.py:176-215 == :344-381 == :543-580)."""
    if net.training:
        return _trunk_train(net, x, bottleneck)
    res = []
    for i, enc in enumerate(net.encoders):
        x = enc(x)
        if net.print_shape:
            print("x_{} : {}".format(i, x.shape))
        res.append(x)

    if bottleneck is not None:
        y = bottleneck(x)
    else:
        # [B,C,1,T] -> [B,T,C] -> projected LSTM -> [B,C,1,T]: one persistent launch per layer for the
        # recurrence (unet_amd/lstm.py); an LSTM of another shape runs torch's device implementation
        seq = x.reshape(x.shape[0], x.shape[1], x.shape[3]).permute(0, 2, 1)
        if lstm_supported(net.bottleneck):
            seq = lstm_forward(net.bottleneck, seq.float(), net._lstm_cache,
                               products=getattr(net, "_lstm_products", "fp32")).to(x.dtype)
        else:
            seq = net.bottleneck(seq)[0]
        y = seq.permute(0, 2, 1).reshape(x.shape[0], x.shape[1], 1, x.shape[3]).contiguous()

    for i, rp in enumerate(net.res_paths):
        res[i] = rp(res[i])

    # ``y = y + res[-1-i]; y = dec(y)`` (ResUNet.py:206-210).  Where decoder i ends in a transposed conv and decoder
    # i+1 starts with a ResBlock, the NEXT add is done in that conv's epilogue (same two fp32 operands, same
    # sum), so every ResBlock layer reads a single tensor; a Decoder-first block reads y and res as two inputs.
    n_dec = len(net.decoders)
    added = False                                      # res[-1-i] is already inside y
    for i, dec in enumerate(net.decoders):
        skip = res[-1 - i]
        if net.print_shape:
            print("y {} += r_{} : {}".format(y.shape, i, skip.shape))
        nxt = net.decoders[i + 1] if i + 1 < n_dec else None
        ends_in_conv = isinstance(dec, Decoder) or not isinstance(dec, ResBlock)
        next_skip = res[-2 - i] if (ends_in_conv and nxt is not None and not isinstance(nxt, Decoder)) else None
        if isinstance(dec, Decoder):
            y = dec(y, skip, add=True, residual=next_skip)       # y + res read as two kernel inputs
        elif isinstance(dec, ResBlock):
            y = dec(y) if added else dec(y, add=skip)
        else:                                          # Sequential(ResBlock, Decoder)
            y = dec[1](dec[0](y) if added else dec[0](y, add=skip), residual=next_skip)
        added = next_skip is not None
        if net.print_shape:
            print("y_{} : {}".format(i, y.shape))
    return y


def _trunk_train(net, x, bottleneck=None):
    """The same trunk in train() mode, as the reference writes it (ResUNet.py:176-210): every block records autograd
    history (convolutions and BatchNorm on the library's kernels, see UNet_m._FusedBlock._run_differentiable and
    ResBlock._forward_train), the skip adds are torch additions, and the projected LSTM is one autograd node on the
    persistent recurrence kernels, forward and backward (unet_amd/lstm.py: lstm_forward_train; an LSTM of another
    shape, or a device the kernels do not fit, runs torch's device LSTM)."""
    res = []
    for enc in net.encoders:
        x = enc(x)
        res.append(x)
    if bottleneck is not None:
        y = bottleneck(x)
    else:
        seq = x.reshape(x.shape[0], x.shape[1], x.shape[3]).permute(0, 2, 1)
        seq = seq.contiguous()
        if trainable_on_device(net.bottleneck, seq):
            seq = lstm_forward_train(net.bottleneck, seq)
        else:
            seq = net.bottleneck(seq)[0]
        y = seq.permute(0, 2, 1).reshape(x.shape[0], x.shape[1], 1, x.shape[3])
    for i, rp in enumerate(net.res_paths):
        res[i] = rp(res[i])
    for i, dec in enumerate(net.decoders):
        y = dec(y + res[-1 - i])
    return y


def _head_train(net, y):
    """The output layer (a bare transposed convolution) under autograd: forward and both gradients on the library's kernels."""
    from . import _lib
    from .train import conv_norm_act
    return conv_norm_act(net.out_layer, None, _lib.ACT_NONE, None, y.contiguous())


class ResUNet(nn.Module):
    def __init__(self, c_in=4, c_out=1, n_fft=512, device="cuda:0", output="mapping", print_shape=False, T=125):
        super().__init__()
        n_hfft = int(n_fft / 2 + 1)
        self.print_shape = print_shape
        self.F, self.T = n_hfft, T
        C = 64
        prelu = dict(activation="PReLU")
        self.layer_data = nn.Sequential(Encoder(c_in - 2, 20, (1, 3), 1, (0, 1), 1), nn.LayerNorm(n_hfft))
        self.layer_AF = nn.Sequential(Encoder(2, C - 20, (1, 3), 1, (0, 1), 1), nn.LayerNorm(n_hfft))

        def down():
            return nn.Sequential(Encoder(C, C, (3, 3), (2, 1), (0, 1), **prelu), ResBlock(C))

        def up(pad, opad):
            return nn.Sequential(ResBlock(C), Decoder(C, C, (4, 1), (2, 1), pad, output_padding=opad, **prelu))

        self.encoders = [ResBlock(C)] + [down() for _ in range(5)] + [
            Encoder(C, 2 * C, (3, 1), (2, 1), (0, 0), **prelu),
            Encoder(2 * C, 6 * C, (3, 1), (1, 1), (0, 0), **prelu)]
        for i, m in enumerate(self.encoders):
            self.add_module("enc_{}".format(i), m)
        self.decoders = [
            Decoder(6 * C, 2 * C, (3, 1), (1, 1), (0, 0), **prelu),
            Decoder(2 * C, C, (4, 1), (2, 1), (1, 0), output_padding=(1, 0), **prelu),
            up((1, 0), (1, 0)), up((1, 0), (1, 0)), up((1, 0), (1, 0)), up((0, 0), (0, 0)), up((1, 0), (1, 0)),
            ResBlock(C)]
        for i, m in enumerate(self.decoders):
            self.add_module("dec_{}".format(i), m)
        self.len_model = len(self.encoders)
        self.res_paths = [Encoder(w, w, 1, 1, 0, 1, **prelu) for w in [C] * 6 + [2 * C, 6 * C]]
        for i, m in enumerate(self.res_paths):
            self.add_module("res_path_{}".format(i), m)
        self.bottleneck = nn.LSTM(6 * C, 10 * C, 3, batch_first=True, proj_size=6 * C)
        self.out_layer = nn.ConvTranspose2d(C, 2, (3, 1), stride=1, padding=(1, 0), dilation=1, output_padding=(0, 0))
        self.last_activation = nn.Tanh()
        self._in_data = self._in_af = self._out = None
        self._lstm_cache = {}

    def _input_branch(self, branch, cache_name, x, out, c_off):
        fused = getattr(self, cache_name)
        if fused is None:
            enc = branch[0]
            fused = FusedConv(enc.conv, enc.bn, enc.acti, swap_hw=True)
            object.__setattr__(self, cache_name, fused)
        ln = branch[1]
        layernorm_f(fused(x), ln.weight.float(), ln.bias.float(), ln.eps, out, c_off)

    # What a bfloat16 input selects.  False (default): bfloat16 at the boundary only -- the input is widened once,
    # every tensor between kernels is fp32 and the output is rounded to bfloat16.  Measured on this network, storing
    # the intermediate tensors as bfloat16 makes the forward slower (36.6 against 30.7 ms at batch 32: its time is
    # in kernels bound by instruction issue, not by HBM bytes, and rows of 501 two-byte elements are not 4-byte
    # aligned, which the 16-byte LDS-DMA groups of the residual-block kernel need) and less accurate (max-norm
    # error 1.0e-2 .. 1.4e-2 against 2e-3).  True: bfloat16 storage of every intermediate tensor (half the
    # activation memory), fp32 arithmetic.
    bf16_storage = False
    # What else a bfloat16 input selects (default path, bf16_storage False): 1 or 2 = the strided Encoder / Decoder
    # convolutions (3x3 stride (2,1), (4,1) / (3,1) transposed: a third of the fp32 forward's time) run with
    # 16-bit OPERANDS (fp16: these layers sit behind BatchNorms) on the matrix cores -- 2: weights as two 16-bit
    # halves -- and so do the LSTM's two time-batched products and the recurrent product of each of its steps (sums,
    # gates, cell and state stay fp32), while every tensor stays fp32 NCHW and the ResBlocks and the 1x1 paths keep
    # fp32 products; 0 = fp32 products everywhere (bfloat16 at the boundary only).
    bf16_operands = 1
    # a non-finite result of that path: unet_amd/blocked.py:fp16_overflow_guard ("defer" | "rerun" | "nan"); the safe
    # operands here are fp32 products (bf16_operands = 0)
    on_fp16_overflow = "defer"

    def forward(self, input):
        require_device_tensor(input, "ResUNet")
        if self.training:
            return self._forward_train(input)
        if input.dtype == torch.bfloat16 and not self.bf16_storage:
            def run(operands):
                blocks = [m for m in self.modules() if isinstance(m, (Encoder, Decoder))] if operands else []
                for m in blocks:
                    m.bf16_operands = int(operands)
                # ... and the LSTM's products (input contributions, projection, the recurrent product of each step)
                # take fp16 operands too; its sums, gates, cell and state stay fp32
                object.__setattr__(self, "_lstm_products", "fp16" if operands else "fp32")
                try:
                    return self.forward(input.float()).to(torch.bfloat16)
                finally:
                    for m in blocks:
                        m.bf16_operands = 0
                    object.__setattr__(self, "_lstm_products", "fp32")
            # fp16 operands do not clip: a tensor beyond +-65504 in front of one of those layers gives infinities
            from .blocked import fp16_overflow_guard
            keep = self.bf16_operands

            def set_safe():
                self.bf16_operands = 0

            def unset_safe():
                self.bf16_operands = keep
            return fp16_overflow_guard(self, run(self.bf16_operands), lambda: run(self.bf16_operands),
                                       active=bool(self.bf16_operands), set_safe=set_safe, unset_safe=unset_safe)
        B, _, F, T = input.shape
        x = torch.empty((B, 64, F, T), dtype=input.dtype, device=input.device)
        self._input_branch(self.layer_data, "_in_data", input[:, :-2].contiguous(), x, 0)
        self._input_branch(self.layer_AF, "_in_af", input[:, -2:].contiguous(), x, 20)

        y = _trunk(self, x)

        if self._out is None:
            object.__setattr__(self, "_out", FusedConv(self.out_layer))
        if y.dtype != torch.float32:
            # bfloat16 storage: the mask logits feed a tanh, so the last (thin) layer and the mask run
            # on widened tensors and only the final product is rounded
            return tanh_mask_mul(self._out(y.float()), input.float().contiguous(), 1e-13).to(y.dtype)
        return tanh_mask_mul(self._out(y), input.contiguous(), 1e-13)


    def _forward_train(self, input):
        """ResUNet.py:161-220 under autograd: the input layers see the [B,C,T,F] view as in the reference (conv +
        BatchNorm + LeakyReLU as one node, LayerNorm(257) as torch's), then the trunk, head and mask."""
        if input.dtype != torch.float32:
            raise NotImplementedError("ResUNet: train mode is fp32 only")
        feature = self.layer_data(input[:, :-2].permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
        af = self.layer_AF(input[:, -2:].permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
        y = _trunk_train(self, torch.cat((feature, af), dim=1))
        mask = self.last_activation(_head_train(self, y)) + 1e-13
        return mask * input[:, :2]


class ResUNetOnFreq(nn.Module):
    """Device ``ResUNetOnFreq`` (ResUNet.py:227-389): the 30-channel variant whose strided
    convolutions are (3,1) and whose forward returns the mask itself.

    Same launches as ``ResUNet`` with one input branch; the 30-channel ResBlocks take the
    depthwise + pointwise launches (the one-launch-per-layer kernel is the 64-channel one) and
    ``nn.LSTM(192, 300, 3, proj_size=192)`` runs on the device-wide-barrier recurrence kernel (hidden size 300
    padded to 304 rows per state slot).  The mask activation is fused into the output layer when the
    epilogue has it (Sigmoid, Softplus at threshold 20); another threshold runs the module.
    """

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
        self.encoders += [nn.Sequential(Encoder(W, W, (3, 1), (2, 1), (0, 0), **kw), ResBlock(30))
                          for _ in range(n_block)]
        self.encoders += [Encoder(W, 64, (3, 1), (2, 1), (0, 0), **kw),
                          Encoder(64, 192, (3, 1), (1, 1), (0, 0), **kw)]
        for i, m in enumerate(self.encoders):
            self.add_module("enc_{}".format(i), m)

        def up(kh):
            return nn.Sequential(ResBlock(W), Decoder(W, W, (kh, 1), (2, 1), (1, 0), output_padding=(1, 0), **kw))

        self.decoders = [Decoder(192, 64, (3, 1), (1, 1), (0, 0), **kw),
                         Decoder(64, W, (4, 1), (2, 1), (1, 0), output_padding=(1, 0), **kw)]
        self.decoders += [up(4) for _ in range(n_block - 2)] + [up(5), up(4), ResBlock(W)]
        for i, m in enumerate(self.decoders):
            self.add_module("dec_{}".format(i), m)
        self.len_model = len(self.encoders)

        self.res_paths = [Encoder(w, w, 1, 1, 0, 1, activation="PReLU") for w in [W] * (n_block + 1) + [64, 192]]
        for i, m in enumerate(self.res_paths):
            self.add_module("res_path_{}".format(i), m)

        self.bottleneck = nn.LSTM(192, 300, 3, batch_first=True, proj_size=192, dropout=dropout)
        self.out_layer = nn.ConvTranspose2d(W, c_out, (3, 1), stride=1, padding=(1, 0), dilation=1,
                                            output_padding=(0, 0))
        if activation == "Softplus":
            self.activation_mask = nn.Softplus(threshold=Softplus_thr)
        elif activation == "Sigmoid":
            self.activation_mask = nn.Sigmoid()
        else:
            self.activation_mask = nn.Softplus()
        self._in = self._out = None
        self._lstm_cache = {}

    def forward(self, input):
        require_device_tensor(input, "ResUNetOnFreq")
        if self.training:                      # ResUNet.py:336-386 under autograd
            if input.dtype != torch.float32:
                raise NotImplementedError("ResUNetOnFreq: train mode is fp32 only")
            x = self.layer_input(input.permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
            return self.activation_mask(_head_train(self, _trunk_train(self, x)))
        B, _, F, T = input.shape
        x = torch.empty((B, 30, F, T), dtype=input.dtype, device=input.device)
        ResUNet._input_branch(self, self.layer_input, "_in", input.contiguous(), x, 0)
        y = _trunk(self, x)
        if self._out is None:
            from .fused import activation_code
            fusable = activation_code(self.activation_mask) is not None
            object.__setattr__(self, "_out", (FusedConv(self.out_layer, None, self.activation_mask if fusable else None),
                                              fusable))
        head, fused_act = self._out
        # the mask comes out of a thin layer: keep it in fp32 whatever the storage type of the trunk
        m = head(y.float() if y.dtype != torch.float32 else y)
        if not fused_act:
            m = self.activation_mask(m)
        return m.to(y.dtype)

    def output(self, mask, feature):
        return mask * feature[:, :2]


class LSTMBlock(nn.Module):
    """[B,C,F',T] -> [B,T,C*F'] -> projected LSTM -> back (UNet_m.py:587-610): the recurrence kernels for hidden
    sizes 300 and 640 (eval, and train() without dropout between the layers), torch's device LSTM otherwise."""

    def __init__(self, n_dim, n_hidden, n_layer=3, proj_size=None, dropout=0.2):
        super().__init__()
        if proj_size is None:
            proj_size = n_dim
        self.rnn = nn.LSTM(n_dim, n_hidden, n_layer, batch_first=True, proj_size=proj_size, dropout=dropout)
        self._lstm_cache = {}

    def forward(self, x):
        require_device_tensor(x, "LSTMBlock")
        B, C, F, T = x.shape
        seq = x.reshape(B, C * F, T).permute(0, 2, 1).float().contiguous()
        if lstm_supported(self.rnn) and not self.training:   # hidden size 300 / 640: one persistent recurrence launch per layer
            y = lstm_forward(self.rnn, seq, self._lstm_cache).to(x.dtype)
        elif self.training and torch.is_grad_enabled() and trainable_on_device(self.rnn, seq):
            y = lstm_forward_train(self.rnn, seq).to(x.dtype)   # the same kernels as an autograd node
        else:
            y = self.rnn(seq)[0].to(x.dtype)
        return y.permute(0, 2, 1).reshape(B, C, F, T).contiguous()


class ResUNetOnFreq2(nn.Module):
    """Device ``ResUNetOnFreq2`` (ResUNet.py:394-591) with the LSTM bottleneck: ResUNetOnFreq's 30-channel
    blocks only, an LSTM over the flattened (channel, frequency) features at 7 frequency rows, ``c_out``-channel
    mask or the MEA head's three masks.

    ``multi_scale=True`` registers the reference's extra modules (``ms_i`` pools, ``ms_block``, ``up_i``) so that
    its checkpoints load with ``strict=True``.  The reference runs the pools and the block beside the encoders
    (ResUNet.py:553-560) but nothing downstream reads their result, so the device forward does not launch them:
    the output is the same with and without the switch.

    The reference indexes the bottleneck's output TENSOR with ``[0]`` (ResUNet.py:565), which drops the batch
    axis: the first sample's bottleneck output is what every sample of the batch receives through the
    following add.  The device path reproduces that."""

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
            raise NotImplementedError("ResUNetOnFreq2: the FTGRU bottleneck fails in the reference's constructor")
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
        self._in = self._out = None

    def _first_sample_bottleneck(self, x):
        y = self.bottleneck(x)[0]                                    # [C, F', T]
        return y.unsqueeze(0).expand(x.shape[0], -1, -1, -1).contiguous()

    def forward(self, input):
        require_device_tensor(input, "ResUNetOnFreq2")
        if self.training:                      # ResUNet.py:530-586 under autograd
            if self.activation == "MEA" or input.dtype != torch.float32:
                raise NotImplementedError("ResUNetOnFreq2: train mode is built for fp32 inputs and the single-mask activations")
            x = self.layer_input(input.permute(0, 1, 3, 2).contiguous()).permute(0, 1, 3, 2)
            # ``self.bottleneck(x)[0]`` on a tensor: sample 0's output, which the skip add then broadcasts over the batch
            y = _trunk_train(self, x, lambda t: self.bottleneck(t)[0])
            return self.activation_mask(_head_train(self, y))
        B, _, F, T = input.shape
        x = torch.empty((B, 30, F, T), dtype=input.dtype, device=input.device)
        ResUNet._input_branch(self, self.layer_input, "_in", input.contiguous(), x, 0)
        y = _trunk(self, x, self._first_sample_bottleneck)
        if self._out is None:
            from .fused import activation_code
            fusable = activation_code(self.activation_mask) is not None
            object.__setattr__(self, "_out", (FusedConv(self.out_layer, None, self.activation_mask if fusable else None),
                                              fusable))
        head, fused_act = self._out
        m = head(y.float() if y.dtype != torch.float32 else y)
        if self.activation == "MEA":
            return tuple(t.to(y.dtype) for t in self.activation_mask(m))
        if not fused_act:
            m = self.activation_mask(m)
        return m.to(y.dtype)

    def output(self, mask, feature):
        if self.activation == "MEA":
            return self.last_activation.output(mask, feature)   # never set: AttributeError, as ResUNet.py:589
        return mask * feature[:, :2]
