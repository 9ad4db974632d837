"""16-bit-operand convolution chain on the channel-blocked layout (csrc/conv_b16.hip).

``BlockedTensor`` carries an activation as ``[N, CB, H, W, 8]`` bfloat16 or float16 (CB = 2 * ceil(C / 16), padded
channels are zero; the tensor's dtype names the format) together with its logical channel count.  A layer's operand
format is its input's: fp16 has three more mantissa bits at the same matrix rate and suits activations of modest
range (behind a BatchNorm); bfloat16 suits a network input of unknown scale.  ``BlockedConv`` is the blocked-layout
counterpart of ``fused.FusedConv``: conv / transposed conv + eval BatchNorm + activation (+ gate, + the
concat with a skip tensor) as one launch of the bf16 MFMA kernel.
"""

import ctypes

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .fused import _pair, activation_code
from .ops import conv_out_size, make_desc, set_launch_observer  # noqa: F401
from . import ops as _ops


blocks_for = _ops.blocks_for


def fp16_overflow_guard(module, out, rerun, active=None, set_safe=None, unset_safe=None):
    """What a network on the fp16-operand chain does with a result that is not finite.  fp16 stores do not clip: an
    activation beyond +-65504 becomes infinity and reaches the output as NaN.  ``module.on_fp16_overflow``:

    "defer" (default)  no host synchronisation: the sum of the result (non-finite exactly when an element is) travels
                       to pinned host memory behind the call, and the NEXT call that finds it arrived reads it.  A
                       non-finite sum raises a warning, switches the module to its safe operands for good
                       (``operand_format = "bf16"``; ResUNet: fp32 products) and that call is recomputed with them
                       -- so the first overflowing call returns its non-finite result, later ones are right.  A
                       check still in flight is not overtaken: calls issued meanwhile are not looked at.
    "rerun"            one host synchronisation per call: a non-finite result is recomputed with the safe operands
                       before it is returned (with a warning); the module keeps its format.
    "nan"              the result is returned as it is.

    ``rerun()`` is the module's own forward on that path (it reads the format from the module); ``set_safe`` /
    ``unset_safe`` switch the module to and from its safe operands (default: ``operand_format``)."""
    mode = getattr(module, "on_fp16_overflow", "defer")
    if mode not in ("defer", "rerun", "nan"):
        raise ValueError("on_fp16_overflow is 'defer', 'rerun' or 'nan'")
    if active is None:
        active = getattr(module, "operand_format", "bf16") == "fp16"
    if mode == "nan" or not active:
        return out
    if set_safe is None:
        def set_safe():
            object.__setattr__(module, "operand_format", "bf16")

        def unset_safe():
            object.__setattr__(module, "operand_format", "fp16")
    import math
    import warnings
    what = ("{}: non-finite values behind the fp16-operand layers (an activation beyond +-65504, or a non-finite "
            "input); ".format(type(module).__name__))
    probe = torch.view_as_real(out) if out.is_complex() else out
    if mode == "rerun":
        if math.isfinite(float(probe.sum(dtype=torch.float32))):
            return out
        warnings.warn(what + "recomputing this call with the safe operands")
        set_safe()
        try:
            return rerun()
        finally:
            unset_safe()
    state = module.__dict__.get("_fp16_check")           # [pinned host scalar, event, in flight]
    if state is None:
        state = [torch.zeros((), dtype=torch.float32, pin_memory=True), torch.cuda.Event(), False]
        object.__setattr__(module, "_fp16_check", state)
    if state[2]:
        if not state[1].query():
            return out                                   # the earlier check has not arrived yet
        state[2] = False
        if not math.isfinite(float(state[0])):
            warnings.warn(what + "an earlier call returned them; this call and the following ones use the safe operands")
            set_safe()
            return rerun()
    state[0].copy_(probe.sum(dtype=torch.float32), non_blocking=True)
    state[1].record()
    state[2] = True
    return out


class BlockedTensor:
    __slots__ = ("data", "C")

    def __init__(self, data, C):
        self.data, self.C = data, C

    @property
    def shape(self):
        N, _, H, W, _ = self.data.shape
        return (N, self.C, H, W)

    @staticmethod
    def from_nchw(x, addend=None, fmt="bf16"):
        """x: contiguous fp32 or bfloat16 device tensor [N,C,H,W]; ``addend`` (same shape and type) is added first
        (in fp32, one rounding); ``fmt``: "bf16" or "fp16", the blocked tensor's 16-bit format."""
        if fmt not in ("bf16", "fp16"):
            raise ValueError("BlockedTensor.from_nchw: fmt is 'bf16' or 'fp16'")
        if addend is None and fmt == "bf16":
            return BlockedTensor(_ops.blocked_from_nchw(x), x.shape[1])
        return BlockedTensor(_ops.blocked_from_nchw_sum(x, addend, fmt == "fp16"), x.shape[1])

    @property
    def fmt(self):
        return "fp16" if self.data.dtype == torch.float16 else "bf16"

    def to_nchw(self, dtype=torch.bfloat16):
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("BlockedTensor.to_nchw: bfloat16 or float32")
        return _ops.nchw_from_blocked(self.data, self.C, dtype == torch.float32)


def _geom_of(conv):
    transposed = isinstance(conv, nn.ConvTranspose2d)
    kh, kw = _pair(conv.kernel_size)
    sh, sw = _pair(conv.stride)
    ph, pw = _pair(conv.padding)
    dh, dw = _pair(conv.dilation)
    oph, opw = _pair(conv.output_padding) if transposed else (0, 0)
    return [conv.out_channels, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw], int(transposed)


_desc = _ops.b16_desc


def pack_b16_weights(weight, C0, C1, geom, transposed, halves=1, f16=False):
    d = _desc(1, C0, C1, 64, 64, geom, transposed, 0, 0, 0, 0, halves, False, f16, False)
    nbytes = _lib.lib.unetmi_conv2d_b16_packed_weight_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _lib.UnetmiError("conv2d_b16_packed_weight_bytes: " + _lib.last_error(), _lib.E_BAD_DESC)
    w = weight.detach().to("cpu", torch.float32).contiguous()
    packed = torch.empty(nbytes // 2, dtype=torch.int16)
    check(_lib.lib.unetmi_conv2d_b16_pack_weights(ctypes.byref(d), w.data_ptr(), packed.data_ptr()),
          "unetmi_conv2d_b16_pack_weights")
    return packed


def conv2d_b16(x, skip, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first=0,
               halves=1, nchw_out=False, out_f16=False):
    """One launch (torch.ops.unetmi.conv2d_b16_fwd).  x, skip: BlockedTensor (skip may be None); returns a
    BlockedTensor (fp16 with ``out_f16``, else bfloat16), or with ``nchw_out`` a plain fp32 [N,K,OH,OW] tensor."""
    out = _ops.conv2d_b16_fwd(x.data, x.C, None if skip is None else skip.data, 0 if skip is None else skip.C, wpacked,
                              bias, scale, shift, act_param, gate, list(geom),
                              [int(transposed), int(act), int(act_first), int(halves), int(nchw_out), int(out_f16)])
    return out if nchw_out else BlockedTensor(out, geom[0])


class BlockedConv:
    """conv [+ BatchNorm2d(eval)] [+ activation] on blocked tensors; parameters are folded and packed on
    first use for the channel split (C0, C1) it is called with, and again when a parameter changes."""

    def __init__(self, conv, norm=None, act=None, halves=1, nchw_out=False, out_fmt="bf16"):
        """``out_fmt``: "bf16" or "fp16", the format of the blocked output (the operands' format is the input's)."""
        if out_fmt not in ("bf16", "fp16"):
            raise ValueError("BlockedConv: out_fmt is 'bf16' or 'fp16'")
        self.out_f16 = out_fmt == "fp16"
        if norm is not None and not isinstance(norm, (nn.BatchNorm2d, nn.Identity)):
            raise ValueError("BlockedConv: only BatchNorm2d (eval) can be folded")
        code = activation_code(act)
        if code is None:
            raise ValueError("BlockedConv: activation {} is not one the epilogue has".format(type(act).__name__))
        self.conv, self.norm = conv, (None if isinstance(norm, nn.Identity) else norm)
        self.act_code, self.act_src = code
        self.halves, self.nchw_out = halves, nchw_out
        self.geom, self.transposed = _geom_of(conv)
        self._key = None
        self._img = {}

    def _versions(self):
        ts = [self.conv.weight, self.conv.bias]
        if self.norm is not None:
            ts += [self.norm.weight, self.norm.bias, self.norm.running_mean, self.norm.running_var]
        if isinstance(self.act_src, torch.Tensor):
            ts.append(self.act_src)
        return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)

    def _prepare(self, C0, C1, device, f16=False):
        key = (self._versions(), str(device), self.norm is not None and self.norm.training)
        if key != self._key:
            self._img.clear()
            self._key = key
        st = self._img.get((C0, C1, f16))
        if st is not None:
            return st
        K = self.geom[0]
        f32 = dict(dtype=torch.float32, device=device)
        bias = self.conv.bias.detach().to(**f32) if self.conv.bias is not None else torch.zeros(K, **f32)
        if self.norm is not None:
            n = self.norm
            if n.training or n.running_mean is None or n.running_var is None:
                raise NotImplementedError("BlockedConv folds BatchNorm2d from its running statistics: the norm must be "
                                          "in eval() mode and track them")
            scale = torch.rsqrt(n.running_var.detach().to(**f32) + n.eps)
            if n.weight is not None:                     # affine=False: no weight / bias
                scale = n.weight.detach().to(**f32) * scale
            shift = -n.running_mean.detach().to(**f32) * scale
            if n.bias is not None:
                shift = shift + n.bias.detach().to(**f32)
        else:
            scale, shift = torch.ones(K, **f32), torch.zeros(K, **f32)
        if isinstance(self.act_src, torch.Tensor):
            ap = self.act_src.detach().to(**f32).reshape(-1).contiguous()
        elif self.act_src is not None:
            ap = torch.tensor([float(self.act_src)], **f32)
        else:
            ap = torch.zeros(1, **f32) if self.act_code == _lib.ACT_LEAKY else None
        st = dict(w=pack_b16_weights(self.conv.weight, C0, C1, self.geom, self.transposed, self.halves, f16).to(device),
                  bias=bias.contiguous(), scale=scale.contiguous(), shift=shift.contiguous(), ap=ap)
        self._img[(C0, C1, f16)] = st
        return st

    def __call__(self, x, skip=None, gate=None, residual=None):
        """``residual`` (fp32 NCHW, shaped like the output; with ``nchw_out`` only) is added to the result."""
        if skip is not None and skip.data.dtype != x.data.dtype:
            raise ValueError("BlockedConv: the two inputs of a layer share one 16-bit format")
        st = self._prepare(x.C, 0 if skip is None else skip.C, x.data.device, x.data.dtype == torch.float16)
        if residual is not None:
            if not self.nchw_out:
                raise ValueError("BlockedConv: a residual goes with the fp32 NCHW output")
            return _ops.conv2d_b16_nchw_fwd(x.data, x.C, None if skip is None else skip.data, 0 if skip is None else skip.C,
                                            st["w"], st["bias"], st["scale"], st["shift"], st["ap"], gate, residual,
                                            list(self.geom), [int(self.transposed), int(self.act_code), 0, int(self.halves)])
        return conv2d_b16(x, skip, st["w"], st["bias"], st["scale"], st["shift"], st["ap"], gate, self.geom,
                          self.transposed, self.act_code, 0, self.halves, self.nchw_out, self.out_f16)

    def with_head(self, x, skip, head_conv, head_act_code, interleaved=True, gate=None):
        """This layer followed by the 1x1 convolution ``head_conv`` (at most four output channels) and its
        activation as ONE launch: the layer's output stays in the fp32 accumulators and only the head's result is
        written -- fp32 [N, OH, OW, P] (``interleaved``; two channels are a complex64 tensor) or [N, P, OH, OW]."""
        if skip is not None and skip.data.dtype != x.data.dtype:
            raise ValueError("BlockedConv: the two inputs of a layer share one 16-bit format")
        st = self._prepare(x.C, 0 if skip is None else skip.C, x.data.device, x.data.dtype == torch.float16)
        # the head's parameters are used as they are (views when they already are fp32 tensors on the device)
        P = head_conv.out_channels
        head_w = head_conv.weight.detach().to(dtype=torch.float32, device=x.data.device).reshape(P, -1).contiguous()
        head_b = None if head_conv.bias is None else \
            head_conv.bias.detach().to(dtype=torch.float32, device=x.data.device).contiguous()
        return _ops.conv2d_b16_head_fwd(x.data, x.C, None if skip is None else skip.data, 0 if skip is None else skip.C,
                                        st["w"], st["bias"], st["scale"], st["shift"], st["ap"], gate, head_w,
                                        head_b, list(self.geom),
                                        [int(self.transposed), int(self.act_code), 0, int(self.halves), int(head_act_code),
                                         int(interleaved)])
