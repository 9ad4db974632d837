"""Host logic of a fused convolution layer.

``FusedConv`` owns nothing but caches: the packed weight image and the folded
epilogue vectors of one (conv | transposed conv) [+ BatchNorm2d] [+ activation]
group, rebuilt whenever a source parameter changes (tracked by tensor version
counters and storage pointers) or the layer is first used on a device.  The
parameters themselves stay in the ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d``
submodules, so ``state_dict`` keys and initialisation are the reference's.
"""

import os

import torch
import torch.nn as nn

from . import _lib
from .ops import conv2d_fwd, pack_conv_weights, pwconv64


# Kernel family requested from the MFMA convolution kernels (reserved[1] of the descriptor): 0 lets the library
# choose from the grid size; tests and A/B timing runs set 1, 2, 4, 5 or 6 (row tiles per wave; there is no 3) to
# run one family on every shape that has it.
ROW_TILES = int(os.environ.get("UNETMI_ROW_TILES", "0"))
if ROW_TILES not in (0, 1, 2, 4, 5, 6):
    raise ImportError("unet_amd: UNETMI_ROW_TILES={} is not a kernel family (0, 1, 2, 4, 5 or 6)".format(ROW_TILES))


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) == 1:
            return int(v[0]), int(v[0])
        return int(v[0]), int(v[1])
    return int(v), int(v)


def activation_code(act):
    """(code, parameter source) for an activation module, or None when the
    module is not one the kernel epilogue implements."""
    if act is None or isinstance(act, nn.Identity):
        return _lib.ACT_NONE, None
    if isinstance(act, nn.LeakyReLU):
        return _lib.ACT_LEAKY, float(act.negative_slope)
    if isinstance(act, nn.PReLU):
        return _lib.ACT_LEAKY, act.weight
    if isinstance(act, nn.ReLU):
        return _lib.ACT_RELU, None
    if isinstance(act, nn.SiLU):
        return _lib.ACT_SILU, None
    if isinstance(act, nn.Softplus) and act.beta == 1 and act.threshold == 20:
        return _lib.ACT_SOFTPLUS, None
    if isinstance(act, nn.Sigmoid):
        return _lib.ACT_SIGMOID, None
    if isinstance(act, nn.Tanh):
        return _lib.ACT_TANH, None
    if isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none":
        return _lib.ACT_GELU, None
    return None


def instance_norm_act(z, norm, act, gate=None):
    """act(norm(z)) [* gate] for an nn.InstanceNorm2d that normalises with each plane's own statistics (its default:
    no running statistics) as ONE launch (unetmi_instnorm_act_fwd), or None when norm / act are not such a pair."""
    if not isinstance(norm, nn.InstanceNorm2d) or (norm.track_running_stats and not norm.training):
        return None
    found = activation_code(act)
    if found is None or z.dim() != 4 or not z.is_contiguous():
        return None
    code, par = found
    if isinstance(par, torch.Tensor):
        ap = par.detach().float().reshape(-1).contiguous()
    elif par is not None:
        ap = torch.full((1,), float(par), dtype=torch.float32, device=z.device)
    else:
        ap = None
    from .ops import instnorm_act
    gamma = norm.weight.detach().float().contiguous() if norm.weight is not None else None
    beta = norm.bias.detach().float().contiguous() if norm.bias is not None else None
    return instnorm_act(z, gamma, beta, norm.eps, code, ap, None if gate is None else gate.float().contiguous())


def layer_norm_act(z, norm, act, gate=None):
    """act(norm(z)) [* gate] for an nn.LayerNorm over the LAST axis of z [N,C,H,W] (what nn.LayerNorm(out_channels)
    does to an Encoder's / Decoder's NCHW output, UNet_m.py:312-319) as ONE launch (unetmi_layernorm_w_act_fwd), or
    None when norm / act / z are not such a case."""
    if not isinstance(norm, nn.LayerNorm) or z.dim() != 4 or not z.is_contiguous():
        return None
    if tuple(norm.normalized_shape) != (z.shape[3],):
        return None
    found = activation_code(act)
    if found is None:
        return None
    code, par = found
    if isinstance(par, torch.Tensor):
        ap = par.detach().float().reshape(-1).contiguous()
    elif par is not None:
        ap = torch.full((1,), float(par), dtype=torch.float32, device=z.device)
    else:
        ap = None
    from .ops import layernorm_w_act
    gamma = norm.weight.detach().float().contiguous() if norm.weight is not None else None
    beta = norm.bias.detach().float().contiguous() if norm.bias is not None else None
    if gamma is None and beta is not None:
        return None
    return layernorm_w_act(z, gamma, beta, norm.eps, code, ap, None if gate is None else gate.float().contiguous())


def require_device_tensor(x, who):
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError(
            "{}: the forward runs only on an MI355X device tensor (got {}); this package has no "
            "CPU path".format(who, "a CPU tensor" if isinstance(x, torch.Tensor) else type(x)))
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("{}: expected a float32 (or bfloat16-storage) input, got {}".format(who, x.dtype))


class FusedConv:
    """conv [+ BatchNorm2d(eval)] [+ activation] as ONE kernel launch."""

    def __init__(self, conv, norm=None, act=None, act_first=False, swap_hw=False):
        """swap_hw: the conv module is defined on [B,C,T,F] (time and frequency
        exchanged, as ResUNet's input layers are, ResUNet.py:165-170) but is applied
        to the unpermuted [B,C,F,T] tensor: kernel, padding, stride and dilation
        swap their two axes and so does the weight."""
        self.conv, self.norm, self.act, self.act_first = conv, norm, act, bool(act_first)
        self.swap_hw = bool(swap_hw)
        self._key = None
        self._state = None

    # -- what can be fused -------------------------------------------------------
    def fusable(self):
        c = self.conv
        if c.groups != 1 or isinstance(c.padding, str):
            return False
        if isinstance(c, nn.Conv2d) and c.padding_mode != "zeros":
            return False
        if self.norm is not None and not isinstance(self.norm, nn.Identity):
            n = self.norm
            if not isinstance(n, nn.BatchNorm2d) or n.training or n.running_mean is None:
                return False
        return activation_code(self.act) is not None

    def geometry(self):
        c = self.conv
        transposed = isinstance(c, nn.ConvTranspose2d)
        kh, kw = _pair(c.kernel_size)
        sh, sw = _pair(c.stride)
        ph, pw = _pair(c.padding)
        dh, dw = _pair(c.dilation)
        oph, opw = _pair(c.output_padding) if transposed else (0, 0)
        if self.swap_hw:
            kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = kw, kh, sw, sh, pw, ph, dw, dh, opw, oph
        return [c.out_channels, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw], int(transposed)

    # -- cache ---------------------------------------------------------------------
    def _sources(self):
        src = [self.conv.weight, self.conv.bias]
        n = self.norm
        if isinstance(n, nn.BatchNorm2d):
            src += [n.weight, n.bias, n.running_mean, n.running_var]
        if isinstance(self.act, nn.PReLU):
            src.append(self.act.weight)
        return src

    def _cache_key(self, device):
        return (str(device), self.act_first, type(self.act).__name__) + tuple(
            (t._version, t.data_ptr()) if t is not None else None for t in self._sources())

    def invalidate(self):
        self._key = None

    def _prepare(self, device):
        key = self._cache_key(device)
        if key == self._key:
            return self._state
        c = self.conv
        geom, transposed = self.geometry()
        K = c.out_channels
        found = activation_code(self.act)
        code, par = found if found is not None else (_lib.ACT_NONE, None)
        with torch.no_grad():
            weight = c.weight.permute(0, 1, 3, 2) if self.swap_hw else c.weight
            wpacked = pack_conv_weights(weight, c.in_channels, 0, geom,
                                        [transposed, _lib.IN_SINGLE, code, 0]).to(device)
            self._weight_for_split, self._split_images = weight.detach().to(torch.float32).contiguous(), {}
            f32 = dict(dtype=torch.float32, device=device)
            bias = (c.bias.detach().to(**f32) if c.bias is not None else torch.zeros(K, **f32)).contiguous()
            n = self.norm
            if isinstance(n, nn.BatchNorm2d):
                inv = torch.rsqrt(n.running_var.detach().to(**f32) + n.eps)
                gamma = n.weight.detach().to(**f32) if n.weight is not None else torch.ones(K, **f32)
                beta = n.bias.detach().to(**f32) if n.bias is not None else torch.zeros(K, **f32)
                scale = (gamma * inv).contiguous()
                shift = (beta - n.running_mean.detach().to(**f32) * scale).contiguous()
            else:
                scale, shift = torch.ones(K, **f32), torch.zeros(K, **f32)
            if isinstance(par, torch.Tensor):
                act_param = par.detach().to(**f32).reshape(-1).contiguous()
                if act_param.numel() not in (1, K):
                    raise ValueError("PReLU with {} parameters on {} channels".format(act_param.numel(), K))
            elif par is not None:
                act_param = torch.full((1,), par, **f32)
            else:
                act_param = None
        self._state = dict(wpacked=wpacked, bias=bias, scale=scale, shift=shift, act_param=act_param,
                           geom=geom, transposed=transposed, act=code)
        self._key = key
        return self._state

    def _is_pointwise_64(self, st, mode):
        c = self.conv
        if st["transposed"] or c.in_channels != 64 or c.out_channels != 64 or mode == _lib.IN_CONCAT:
            return False
        if st["geom"][1:9] != [1, 1, 1, 1, 0, 0, 1, 1] or st["act"] not in (_lib.ACT_NONE, _lib.ACT_LEAKY):
            return False
        return st["act_param"] is None or st["act_param"].numel() == 1

    def _image(self, st, C0, C1, mode):
        """Packed weights for this call: a concatenated input is chunked per source inside the
        kernel, so its image depends on where the channels split (cached per split)."""
        if mode != _lib.IN_CONCAT:
            return st["wpacked"]
        img = self._split_images.get((C0, C1))
        if img is None:
            img = pack_conv_weights(self._weight_for_split, C0, C1, st["geom"],
                                    [st["transposed"], _lib.IN_CONCAT, st["act"], 0]).to(st["wpacked"].device)
            self._split_images[(C0, C1)] = img
        return img

    # -- launch ----------------------------------------------------------------------
    def __call__(self, x, x2=None, in_mode=None, gate=None, residual=None, epilogue=True):
        """x2 with in_mode IN_CONCAT (default when x2 is given) or IN_ADD.
        epilogue=False runs the bare convolution + bias (used when the norm or
        activation around it is not one the kernel fuses)."""
        st = self._prepare(x.device)
        if x2 is None:
            mode = _lib.IN_SINGLE
        else:
            mode = _lib.IN_CONCAT if in_mode is None else in_mode
        cin = x.shape[1] + (x2.shape[1] if (x2 is not None and mode == _lib.IN_CONCAT) else 0)
        if cin != self.conv.in_channels:
            raise RuntimeError("expected {} input channels, got {}".format(self.conv.in_channels, cin))
        if epilogue and residual is None and gate is None and self._is_pointwise_64(st, mode):
            # 64 -> 64 pointwise layer: the bandwidth-bound one-pass kernel
            slope = st["act_param"] if st["act"] == _lib.ACT_LEAKY else None
            return pwconv64(x, x2, self._weight_for_split, st["bias"], st["scale"], st["shift"], slope,
                            self.act_first, False)
        wimg = self._image(st, x.shape[1], x2.shape[1] if x2 is not None else 0, mode)
        if epilogue:
            try:
                return conv2d_fwd(x, x2, wimg, st["bias"], st["scale"], st["shift"],
                                  st["act_param"], gate, residual, st["geom"],
                                  [st["transposed"], mode, st["act"], int(self.act_first), ROW_TILES])
            except _lib.UnetmiError as e:
                if e.code != _lib.E_UNSUPPORTED or (x2 is None and x.dtype == torch.float32):
                    raise
            if x.dtype != torch.float32:
                # a layer shape the bfloat16-storage kernels do not cover (time-stride-2 transposed
                # convolutions): widen the operands, run the fp32 kernel, round the result
                wide = self(x.float(), None if x2 is None else x2.float(), in_mode=in_mode, gate=gate,
                            residual=None if residual is None else residual.float())
                return wide.to(x.dtype)
            # the one case the kernels do not take two sources for (a summed input on a
            # time-stride-2 transposed conv): form the combined input, run the single-source kernel
            merged = torch.cat([x, x2], dim=1) if mode == _lib.IN_CONCAT else x + x2
            return conv2d_fwd(merged, None, st["wpacked"], st["bias"], st["scale"], st["shift"],
                              st["act_param"], gate, residual, st["geom"],
                              [st["transposed"], _lib.IN_SINGLE, st["act"], int(self.act_first), ROW_TILES])
        ones = torch.ones_like(st["bias"])
        return conv2d_fwd(x, x2, wimg, st["bias"], ones, torch.zeros_like(ones), None, gate,
                          residual, st["geom"], [st["transposed"], mode, _lib.ACT_NONE, 0, ROW_TILES])
