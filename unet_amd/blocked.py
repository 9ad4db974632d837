"""bfloat16-operand convolution chain on the channel-blocked layout (csrc/conv_b16.hip).

``BlockedTensor`` carries an activation as ``[N, CB, H, W, 8]`` bfloat16 (CB = 2 * ceil(C / 16), padded
channels are zero) together with its logical channel count.  ``BlockedConv`` is the blocked-layout
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


def blocks_for(C):
    return 2 * ((C + 15) // 16)


class BlockedTensor:
    __slots__ = ("data", "C")

    def __init__(self, data, C):
        self.data, self.C = data, C

    @property
    def shape(self):
        N, _, H, W, _ = self.data.shape
        return (N, self.C, H, W)

    @staticmethod
    def from_nchw(x):
        """x: contiguous fp32 or bfloat16 device tensor [N,C,H,W]."""
        if not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
            raise ValueError("BlockedTensor.from_nchw: contiguous 4-D device tensor expected")
        N, C, H, W = x.shape
        out = torch.empty((N, blocks_for(C), H, W, 8), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            check(_lib.lib.unetmi_blocked_from_nchw(N, C, H * W, x.data_ptr(), _ops.dtype_code(x), out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "unetmi_blocked_from_nchw")
        return BlockedTensor(out, C)

    def to_nchw(self, dtype=torch.bfloat16):
        N, C, H, W = self.shape
        out = torch.empty((N, C, H, W), dtype=dtype, device=self.data.device)
        code = _lib.DTYPE_BF16 if dtype == torch.bfloat16 else _lib.DTYPE_F32
        with torch.cuda.device(out.device):
            check(_lib.lib.unetmi_nchw_from_blocked(N, C, H * W, self.data.data_ptr(), out.data_ptr(), code,
                                                    torch.cuda.current_stream().cuda_stream), "unetmi_nchw_from_blocked")
        return out


def _geom_of(conv):
    transposed = isinstance(conv, nn.ConvTranspose2d)
    kh, kw = _pair(conv.kernel_size)
    sh, sw = _pair(conv.stride)
    ph, pw = _pair(conv.padding)
    dh, dw = _pair(conv.dilation)
    oph, opw = _pair(conv.output_padding) if transposed else (0, 0)
    return [conv.out_channels, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw], int(transposed)


def _desc(N, C0, C1, H, W, geom, transposed, act, act_first, act_pc, has_gate, halves, nchw_out):
    d = make_desc(N, C0, C1, H, W, geom, [transposed, _lib.IN_CONCAT if C1 else _lib.IN_SINGLE, act, act_first],
                  act_pc, has_gate, 0, _lib.DTYPE_BF16)
    d.reserved[1] = halves
    d.reserved[2] = int(nchw_out)
    return d


def pack_b16_weights(weight, C0, C1, geom, transposed, halves=1):
    d = _desc(1, C0, C1, 64, 64, geom, transposed, 0, 0, 0, 0, halves, False)
    nbytes = _lib.lib.unetmi_conv2d_b16_packed_weight_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _lib.UnetmiError("conv2d_b16_packed_weight_bytes: " + _lib.last_error(), _lib.E_BAD_DESC)
    w = weight.detach().to("cpu", torch.float32).contiguous()
    packed = torch.empty(nbytes // 2, dtype=torch.int16)
    check(_lib.lib.unetmi_conv2d_b16_pack_weights(ctypes.byref(d), w.data_ptr(), packed.data_ptr()),
          "unetmi_conv2d_b16_pack_weights")
    return packed


def conv2d_b16(x, skip, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first=0,
               halves=1, nchw_out=False):
    """One launch.  x, skip: BlockedTensor (skip may be None); returns a BlockedTensor, or with
    ``nchw_out`` a plain fp32 [N,K,OH,OW] tensor."""
    N, C0, H, W = x.shape
    C1 = 0
    if skip is not None:
        if skip.data.shape[0] != N or tuple(skip.data.shape[2:4]) != (H, W):
            raise ValueError("conv2d_b16: skip {} does not match x {}".format(skip.shape, x.shape))
        C1 = skip.C
    K = geom[0]
    act_pc = 1 if (act_param is not None and act_param.numel() > 1) else 0
    d = _desc(N, C0, C1, H, W, geom, transposed, act, act_first, act_pc, int(gate is not None), halves, nchw_out)
    expect = _lib.lib.unetmi_conv2d_b16_packed_weight_bytes(ctypes.byref(d))
    if expect == 0:
        raise _lib.UnetmiError("conv2d_b16: " + _lib.last_error(), _lib.E_BAD_DESC)
    if wpacked.numel() * wpacked.element_size() != expect:
        raise ValueError("conv2d_b16: packed weight image has {} bytes, expected {}".format(
            wpacked.numel() * wpacked.element_size(), expect))
    for t, name in ((bias, "bias"), (scale, "scale"), (shift, "shift")):
        if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (K,) or not t.is_contiguous():
            raise ValueError("conv2d_b16: {} must be a contiguous fp32 device tensor [K]".format(name))
    if gate is not None and (tuple(gate.shape) != (N, K) or gate.dtype != torch.float32 or not gate.is_contiguous()):
        raise ValueError("conv2d_b16: gate must be a contiguous fp32 [N,K] tensor")
    dev = x.data.device
    if nchw_out:
        out = torch.empty((N, K, d.OH, d.OW), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((N, blocks_for(K), d.OH, d.OW, 8), dtype=torch.bfloat16, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        obs = _ops._launch_observer
        if obs is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        check(_lib.lib.unetmi_conv2d_b16_fwd(ctypes.byref(d), x.data.data_ptr(), None if skip is None else skip.data.data_ptr(),
                                             wpacked.data_ptr(), bias.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                             None if act_param is None else act_param.data_ptr(),
                                             None if gate is None else gate.data_ptr(), out.data_ptr(), stream),
              "unetmi_conv2d_b16_fwd")
        if obs is not None:
            ev1.record()
            obs("conv2d_b16_fwd", d, ev0, ev1)
    return out if nchw_out else BlockedTensor(out, K)


class BlockedConv:
    """conv [+ BatchNorm2d(eval)] [+ activation] on blocked tensors; parameters are folded and packed on
    first use for the channel split (C0, C1) it is called with, and again when a parameter changes."""

    def __init__(self, conv, norm=None, act=None, halves=1, nchw_out=False):
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

    def _prepare(self, C0, C1, device):
        key = (self._versions(), str(device))
        if key != self._key:
            self._img.clear()
            self._key = key
        st = self._img.get((C0, C1))
        if st is not None:
            return st
        K = self.geom[0]
        f32 = dict(dtype=torch.float32, device=device)
        bias = self.conv.bias.detach().to(**f32) if self.conv.bias is not None else torch.zeros(K, **f32)
        if self.norm is not None:
            n = self.norm
            scale = (n.weight.detach().to(**f32) / torch.sqrt(n.running_var.detach().to(**f32) + n.eps))
            shift = n.bias.detach().to(**f32) - n.running_mean.detach().to(**f32) * scale
        else:
            scale, shift = torch.ones(K, **f32), torch.zeros(K, **f32)
        if isinstance(self.act_src, torch.Tensor):
            ap = self.act_src.detach().to(**f32).reshape(-1).contiguous()
        elif self.act_src is not None:
            ap = torch.tensor([float(self.act_src)], **f32)
        else:
            ap = torch.zeros(1, **f32) if self.act_code == _lib.ACT_LEAKY else None
        st = dict(w=pack_b16_weights(self.conv.weight, C0, C1, self.geom, self.transposed, self.halves).to(device),
                  bias=bias.contiguous(), scale=scale.contiguous(), shift=shift.contiguous(), ap=ap)
        self._img[(C0, C1)] = st
        return st

    def __call__(self, x, skip=None, gate=None):
        st = self._prepare(x.C, 0 if skip is None else skip.C, x.data.device)
        return conv2d_b16(x, skip, st["w"], st["bias"], st["scale"], st["shift"], st["ap"], gate, self.geom,
                          self.transposed, self.act_code, 0, self.halves, self.nchw_out)
