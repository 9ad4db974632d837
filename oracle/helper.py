"""CPU restatement of the reference's wave-in / wave-out wrapper ``DSUNet_helper``
(DSUNet.py:634-790).  TEST INFRASTRUCTURE ONLY, like the rest of this package.

Kept faithful to the reference, including two things a reader might take for slips:
``short = 16 - T % 16`` pads a whole extra 16 frames when T is already a multiple of 16
(DSUNet.py:695), and the source-to-microphone distance is ``torch.norm`` over the WHOLE
[B,3] difference (DSUNet.py:748), so every sample of a batch shares one set of delays.
"""

import torch
import torch.nn as nn

from .nets import DSUNet


class DSUNet_helper(nn.Module):
    def __init__(self, architecture, n_fft=512, dropout=0.0, bottleneck="None", model_complexity=45, use_SV=True,
                 mag_phase=False, corr=False, DSB=False, activation="LeakyReLU", type_norm="BatchNorm2d",
                 type_masking="CRM", type_encoder="Encoder", type_residual="None"):
        super().__init__()
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

    # DSUNet.py:689-729
    def forward(self, x, angle, mic_pos):
        B, C, L = x.shape
        x = torch.reshape(x, (B * C, L))
        X = torch.stft(x, n_fft=self.n_fft, window=self.window.to(x.device), return_complex=True)
        _, F, T = X.shape
        short = 16 - T % 16
        X = torch.nn.functional.pad(X, (0, short))
        _, F, T = X.shape
        X = X.reshape(B, C, F, T)
        if self.corr:
            for i in range(B):
                for j in range(C - 1):
                    X[i, j + 1] = X[i, 0] * X[i, j + 1]
        SV = self.steering_vector(angle, mic_pos)
        SV = torch.cat((SV.real, SV.imag), -1)
        theta = self.anlge_pre(angle)
        spectral_feature = torch.cat((X.real, X.imag), dim=1)
        if self.DSB:
            dsb = self.delay_n_sum(X, angle, mic_pos)
            dsb = torch.stack((dsb.real, dsb.imag), dim=1)
            spectral_feature = torch.cat((spectral_feature, dsb), dim=1)
        mask = self.net(spectral_feature, SV, theta)
        Y = self.net.masking(X[:, 0], mask)
        return torch.istft(Y, n_fft=self.n_fft, window=self.window.to(Y.device), length=L)

    # DSUNet.py:731-734 (the method name is the reference's spelling)
    def anlge_pre(self, angle):
        sin_theta = torch.sin((-angle) / 180 * torch.pi)
        cos_theta = torch.cos((-angle) / 180 * torch.pi)
        return torch.stack((sin_theta, cos_theta), 1)

    def _tdoa(self, angle, mic_pos):       # DSUNet.py:741-750, :770-779
        B = angle.shape[0]
        loc_src = torch.zeros(B, 3).to(mic_pos.device)
        loc_src[:, 0] = self.dist * torch.cos((-angle) / 180 * torch.pi) * torch.sin(torch.tensor(90 / 180 * torch.pi))
        loc_src[:, 1] = self.dist * torch.sin((-angle) / 180 * torch.pi) * torch.sin(torch.tensor(90 / 180 * torch.pi))
        loc_src[:, 2] = self.dist * torch.cos(torch.tensor((90) / 180 * torch.pi))
        TDOA = torch.zeros(B, self.n_channel).to(mic_pos.device)
        for i in range(self.n_channel):
            TDOA[:, i] = torch.norm(mic_pos[:, i] - loc_src)
        return TDOA[:, :] - TDOA[:, 0:1]

    def _phasors(self, angle, mic_pos, sign):
        B = angle.shape[0]
        TDOA = self._tdoa(angle, mic_pos)
        const = sign * 1j * 2 * torch.pi * self.sr / (self.n_fft * self.ss)
        out = torch.zeros(B, self.n_channel, self.n_hfft, dtype=torch.cfloat).to(mic_pos.device)
        for i in torch.arange(B):
            for j in range(self.n_hfft):
                out[i, :, j] = torch.exp(j * TDOA[i] * const)
                out[i, :, j] /= torch.norm(out[i, :, j])
        return out

    def steering_vector(self, angle, mic_pos):       # DSUNet.py:736-760
        return self._phasors(angle, mic_pos, -1)

    def delay_n_sum(self, X, angle, mic_pos):         # DSUNet.py:763-790
        h = torch.unsqueeze(self._phasors(angle, mic_pos, +1), -1)
        return torch.einsum("bcft,bcfl->bft", [X, h])


class UDSS_helper(DSUNet_helper):
    """Wave-in / wave-out wrapper of the complex-valued network (UDSS.py:685-839).  Same transforms, steering
    vector and delay-and-sum as ``DSUNet_helper``; the spectral feature is ``[B, C, F, T, 2]`` (real and
    imaginary parts stacked last, UDSS.py:761) and the beamformer output is appended as one complex channel
    (UDSS.py:763-772)."""

    def __init__(self, n_fft=512, complex=True, dropout=0.0, bottleneck="None", model_complexity=45,
                 type_encoder="Complex", use_SV=True, mag_phase=False, corr=False, DSB=False, activation="LeakyReLU",
                 type_norm="BatchNorm2d", type_masking="CRM"):
        nn.Module.__init__(self)
        from .udss import UDSS
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

    def forward(self, x, angle, mic_pos):
        B, C, L = x.shape
        x = torch.reshape(x, (B * C, L))
        X = torch.stft(x, n_fft=self.n_fft, window=self.window.to(x.device), return_complex=True)
        _, F, T = X.shape
        short = 16 - T % 16
        X = torch.nn.functional.pad(X, (0, short))
        _, F, T = X.shape
        X = X.reshape(B, C, F, T)
        if self.corr:
            for i in range(B):
                for j in range(C - 1):
                    X[i, j + 1] = X[i, 0] * X[i, j + 1]
        SV = self.steering_vector(angle, mic_pos)
        SV = torch.cat((SV.real, SV.imag), -1)
        theta = self.anlge_pre(angle)
        spectral_feature = torch.stack((X.real, X.imag), dim=-1)
        if self.DSB:
            dsb = self.delay_n_sum(X, angle, mic_pos)
            dsb = torch.unsqueeze(torch.stack((dsb.real, dsb.imag), dim=-1), 1)
            spectral_feature = torch.cat((spectral_feature, dsb), dim=1)
        mask = self.net(spectral_feature, SV, theta)
        Y = self._apply_mask(X[:, 0], mask)
        return torch.istft(Y, n_fft=self.n_fft, window=self.window.to(Y.device), length=L)

    def _apply_mask(self, X0, mask):
        return self.net.masking(X0, mask)


class DAU_helper(UDSS_helper):
    """Wave-in / wave-out wrapper of ``DAU`` (DTU.py:479-629): the transforms, steering vector, delay-and-sum and
    feature layout of ``UDSS_helper``; the mask multiplies the reference channel directly (DTU.py:562)."""

    def __init__(self, n_fft=512, complex=True, dropout=0.0, bottleneck="None", model_complexity=45,
                 type_encoder="Complex", use_SV=True, mag_phase=False, corr=False, DSB=False, activation="LeakyReLU"):
        nn.Module.__init__(self)
        from .udss import DAU
        self.n_fft = n_fft
        self.n_hfft = n_fft // 2 + 1
        self.n_channel = 4
        model_channel = self.n_channel + (1 if DSB else 0)
        self.complex, self.mag_phase, self.use_SV, self.corr, self.DSB = complex, mag_phase, use_SV, corr, DSB
        self.net = DAU(input_channels=model_channel, complex=complex, dropout=dropout, bottleneck=bottleneck,
                       model_complexity=model_complexity, type_encoder=type_encoder, activation=activation)
        self.sr, self.ss, self.dist = 16000, 340.4, 100
        self.window = torch.hann_window(self.n_fft)

    def _apply_mask(self, X0, mask):
        return X0 * mask
