"""``DAU`` / ``DAU_helper``: the directional attractor U-Net of DTU.py:227-629 on the device.

The reference's DAU is the earlier form of UDSS: the same ten complex encoders gated by the attractor code, ten
complex decoders on [decoder, skip] concatenations and complex 1x1 head (same layer table, DTU.py:380-477), with
BatchNorm only, no bottleneck, and the mask always ``tanh`` of the head (DTU.py:355-358).  So the device forward
is ``unet_amd.UDSS``'s -- every complex layer one fused launch on stacked real/imaginary channels -- behind the
reference's constructor, attribute names and ``state_dict`` keys (``masking`` where UDSS has ``mask``).
"""

import torch
import torch.nn as nn

from .UDSS import UDSS, UDSS_helper
from .UNet_m import MEA
from .fused import require_device_tensor


class _DAUCore(UDSS):
    _announce = "DAU"


class DAU(nn.Module):
    """DTU.py:227-378.  ``type_masking`` ("CRM" / "MEA") only selects the ``masking`` member, which the
    reference's forward never calls; bottleneck "None" only, as there."""

    def __init__(self, input_channels=4, n_fft=512, complex=True, model_complexity=45, bottleneck="None",
                 padding_mode="zeros", type_encoder="Complex", type_masking="CRM", activation="LeakyReLU",
                 dropout=0.0):
        super().__init__()
        if bottleneck != "None":
            raise Exception("Not Implemented")
        if type_masking not in ("CRM", "MEA"):
            raise Exception("Not Implemented")
        if activation == "PReLU":        # DTU.py:181 passes inplace=True to nn.PReLU, which has no such argument
            raise TypeError("PReLU.__init__() got an unexpected keyword argument 'inplace'")
        core = _DAUCore(input_channels=input_channels, n_fft=n_fft, complex=complex, model_complexity=model_complexity,
                        bottleneck="None", padding_mode=padding_mode, type_encoder=type_encoder, type_masking="CRM",
                        activation=activation, dropout=dropout, type_norm="BatchNorm2d")
        # the core is not a registered child: its layers are adopted under the reference's own names, so the
        # parameters appear once, with DAU's state_dict keys
        object.__setattr__(self, "_core", core)
        for name in ("bottleneck", "complex", "n_channel", "model_length", "dropout", "padding_mode", "enc_channels",
                     "enc_kernel_sizes", "enc_strides", "enc_paddings", "dec_channels", "dec_kernel_sizes",
                     "dec_strides", "dec_paddings", "dec_output_paddings", "attractEncoders"):
            setattr(self, name, getattr(core, name))
        for name, child in core.named_children():
            if name == "mask":
                self.add_module("masking", nn.Identity() if type_masking == "CRM" else MEA())
            else:
                self.add_module(name, child)

    def train(self, mode=True):
        super().train(mode)
        self._core.training = mode            # the shared layers were switched by the call above
        return self

    @property
    def bf16_matrix(self):
        return self._core.bf16_matrix

    @bf16_matrix.setter
    def bf16_matrix(self, value):
        self._core.bf16_matrix = bool(value)

    def set_size(self, model_complexity, model_depth=20, input_channels=1):
        self._core.set_size(model_complexity, model_depth, input_channels)

    def forward(self, sf, SV, theta):
        require_device_tensor(sf, "DAU")
        return self._core(sf, SV, theta)          # train() mode: UDSS._forward_train, every block under autograd


class DAU_helper(UDSS_helper):
    """Wave in, wave out around ``DAU`` with the reference's constructor and ``forward(x, angle, mic_pos)``
    (DTU.py:479-629); the mask multiplies the reference channel's spectrum (DTU.py:562)."""

    def __init__(self, n_fft=512, complex=True, dropout=0.0, bottleneck="None", model_complexity=45,
                 type_encoder="Complex", use_SV=True, mag_phase=False, corr=False, DSB=False, activation="LeakyReLU"):
        nn.Module.__init__(self)
        from .stft import DeviceSTFT
        self.n_fft = n_fft
        self.n_hfft = n_fft // 2 + 1
        self.n_channel = 4
        model_channel = self.n_channel + (1 if DSB else 0)
        self.complex, self.mag_phase, self.use_SV, self.corr, self.DSB = complex, mag_phase, use_SV, corr, DSB
        self.net = DAU(input_channels=model_channel, complex=complex, dropout=dropout, bottleneck=bottleneck,
                       model_complexity=model_complexity, type_encoder=type_encoder, activation=activation)
        self.sr, self.ss, self.dist = 16000, 340.4, 100
        self.window = torch.hann_window(self.n_fft)
        self._stft = DeviceSTFT(n_fft, self.window)

    def _apply_mask(self, X0, mask):
        return X0 * mask
