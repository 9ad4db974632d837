"""torch.library registration of the libunetmi.so entry points.

``torch.ops.unetmi.conv2d_fwd`` is the fused convolution layer
(include/unetmi.h: unetmi_conv2d_fwd).  Its only implementation is the HIP
kernel, registered for the CUDA dispatch key (ROCm tensors dispatch there);
CPU tensors have no kernel and raise.  ``register_fake`` gives shape inference.
"""

import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check

_LIBRARY = torch.library.Library("unetmi", "DEF")
_LIBRARY.define(
    "conv2d_fwd(Tensor in0, Tensor? in1, Tensor wpacked, Tensor bias, Tensor scale, Tensor shift, "
    "Tensor? act_param, Tensor? gate, Tensor? residual, int[] geom, int[] flags) -> Tensor")

# geom  = [K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw]
# flags = [transposed, in_mode, act, act_first]


def conv_out_size(size, k, s, p, d, op, transposed):
    if transposed:
        return (size - 1) * s - 2 * p + d * (k - 1) + op + 1
    return (size + 2 * p - d * (k - 1) - 1) // s + 1


def dtype_code(t):
    """UNETMI_DTYPE_* of an activation tensor (fp32 or bfloat16 storage)."""
    if t.dtype == torch.float32:
        return _lib.DTYPE_F32
    if t.dtype == torch.bfloat16:
        return _lib.DTYPE_BF16
    raise ValueError("activations must be float32 or bfloat16, got {}".format(t.dtype))


def make_desc(N, C0, C1, H, W, geom, flags, act_pc=0, has_gate=0, has_residual=0, dtype=0):
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    transposed, in_mode, act, act_first = flags
    d = ConvDesc()
    d.N, d.C0, d.C1, d.H, d.W, d.K = N, C0, C1, H, W, K
    d.OH = conv_out_size(H, kh, sh, ph, dh, oph, transposed)
    d.OW = conv_out_size(W, kw, sw, pw, dw, opw, transposed)
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw, d.oph, d.opw = kh, kw, sh, sw, ph, pw, dh, dw, oph, opw
    d.transposed, d.in_mode, d.act, d.act_first = transposed, in_mode, act, act_first
    d.act_param_per_channel, d.has_gate, d.has_residual = act_pc, has_gate, has_residual
    d.reserved[0] = dtype
    return d


def pack_conv_weights(weight, C0, C1, geom, flags):
    """Host-side repack of a torch-layout weight into the kernel's image
    (unetmi_conv2d_pack_weights).  Returns a CPU float32 tensor."""
    flags = list(flags)
    flags[1] = _lib.IN_CONCAT if C1 else _lib.IN_SINGLE
    d = make_desc(1, C0, C1, 64, 64, geom, flags)
    # H/W do not enter the image; give the descriptor any consistent sizes
    d.H = d.W = 64
    d.OH = conv_out_size(64, d.kh, d.sh, d.ph, d.dh, d.oph, d.transposed)
    d.OW = conv_out_size(64, d.kw, d.sw, d.pw, d.dw, d.opw, d.transposed)
    nbytes = _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise _lib.UnetmiError("packed_weight_bytes: " + _lib.last_error())
    w = weight.detach().to("cpu", torch.float32).contiguous()
    packed = torch.empty(nbytes // 4, dtype=torch.float32)
    check(_lib.lib.unetmi_conv2d_pack_weights(ctypes.byref(d), w.data_ptr(), packed.data_ptr()),
          "unetmi_conv2d_pack_weights")
    return packed


# Optional per-launch observer (bench.py's roofline leg): called as
# fn(op_name, desc, start_event, end_event) with torch.cuda events recorded on
# the launch stream immediately around the kernel.
_launch_observer = None


def set_launch_observer(fn):
    global _launch_observer
    _launch_observer = fn


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need(t, name, shape=None, dtype=torch.float32):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("unetmi::conv2d_fwd: {} must be a contiguous {} device tensor".format(name, dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("unetmi::conv2d_fwd: {} has shape {}, expected {}".format(
            name, tuple(t.shape), tuple(shape)))


def _conv2d_fwd_cuda(in0, in1, wpacked, bias, scale, shift, act_param, gate, residual, geom, flags):
    code = dtype_code(in0)
    _need(in0, "in0", dtype=in0.dtype)
    N, C0, H, W = in0.shape
    C1 = 0
    if in1 is not None:
        _need(in1, "in1", dtype=in0.dtype)
        if in1.shape[0] != N or tuple(in1.shape[2:]) != (H, W):
            raise ValueError("unetmi::conv2d_fwd: in1 {} does not match in0 {}".format(
                tuple(in1.shape), tuple(in0.shape)))
        C1 = in1.shape[1]
    K = geom[0]
    act_pc = 1 if (act_param is not None and act_param.numel() > 1) else 0
    d = make_desc(N, C0, C1, H, W, geom, flags, act_pc, int(gate is not None), int(residual is not None), code)
    for t, name in ((bias, "bias"), (scale, "scale"), (shift, "shift")):
        _need(t, name, (K,))
    _need(wpacked, "wpacked")
    if act_param is not None:
        _need(act_param, "act_param", (K,) if act_pc else (1,))
    out = torch.empty((N, K, d.OH, d.OW), dtype=in0.dtype, device=in0.device)
    if gate is not None:
        _need(gate, "gate", (N, K))
    if residual is not None:
        _need(residual, "residual", out.shape, dtype=in0.dtype)
    expect = _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d))
    if expect == 0:
        raise _lib.UnetmiError("unetmi::conv2d_fwd: " + _lib.last_error())
    if wpacked.numel() * 4 != expect:
        raise ValueError("unetmi::conv2d_fwd: packed weight image has {} bytes, expected {}".format(
            wpacked.numel() * 4, expect))
    with torch.cuda.device(in0.device):
        stream = torch.cuda.current_stream().cuda_stream
        obs = _launch_observer
        if obs is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        check(_lib.lib.unetmi_conv2d_fwd(ctypes.byref(d), _ptr(in0), _ptr(in1), _ptr(wpacked),
                                         _ptr(bias), _ptr(scale), _ptr(shift), _ptr(act_param),
                                         _ptr(gate), _ptr(residual), _ptr(out), stream),
              "unetmi_conv2d_fwd")
        if obs is not None:
            ev1.record()
            obs("conv2d_fwd", d, ev0, ev1)
    return out


_LIBRARY.impl("conv2d_fwd", _conv2d_fwd_cuda, "CUDA")


@torch.library.register_fake("unetmi::conv2d_fwd")
def _conv2d_fwd_fake(in0, in1, wpacked, bias, scale, shift, act_param, gate, residual, geom, flags):
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    N, _, H, W = in0.shape
    return in0.new_empty((N, K, conv_out_size(H, kh, sh, ph, dh, oph, flags[0]),
                          conv_out_size(W, kw, sw, pw, dw, opw, flags[0])))


conv2d_fwd = torch.ops.unetmi.conv2d_fwd


# ---------------------------------------------------------------------------------------
# bandwidth-bound ops (csrc/pointwise.hip)
# ---------------------------------------------------------------------------------------
_LIBRARY.define("dwconv2d_fwd(Tensor x, Tensor? x2, Tensor weight, Tensor? bias, int[] padding, int[] dilation) -> Tensor")
_LIBRARY.define("layernorm_f(Tensor x, Tensor gamma, Tensor beta, float eps, Tensor(a!) out, int c_off) -> ()")
_LIBRARY.define("tanh_mask_mul(Tensor y, Tensor x, float eps) -> Tensor")
_LIBRARY.define("crf_apply(Tensor H, Tensor X) -> Tensor")
_LIBRARY.define("glu(Tensor z, Tensor? scale, Tensor? shift, int act, Tensor? act_param, Tensor? gate) -> Tensor")
_LIBRARY.define("pwconv64(Tensor x, Tensor? x2, Tensor weight, Tensor? bias, Tensor scale, Tensor shift, Tensor? slope, "
                "bool act_first, bool residual) -> Tensor")
_LIBRARY.define("resblock_layer(Tensor x, Tensor? x2, Tensor dw_weight, Tensor? dw_bias, Tensor pw_weight, "
                "Tensor? pw_bias, Tensor alpha, Tensor scale, Tensor shift, int dilation, bool residual) -> Tensor")


def _dev(t, name, op, dtype=torch.float32):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("unetmi::{}: {} must be a contiguous {} device tensor".format(op, name, dtype))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _dwconv2d_fwd_cuda(x, x2, weight, bias, padding, dilation):
    code = dtype_code(x)
    _dev(x, "x", "dwconv2d_fwd", x.dtype), _dev(weight, "weight", "dwconv2d_fwd")
    N, C, H, W = x.shape
    if tuple(weight.shape[:2]) != (C, 1):
        raise ValueError("unetmi::dwconv2d_fwd: weight must be [C,1,kh,kw]")
    if x2 is not None:
        _dev(x2, "x2", "dwconv2d_fwd", x.dtype)
        if x2.shape != x.shape:
            raise ValueError("unetmi::dwconv2d_fwd: x2 shape differs from x")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_dwconv2d_fwd(N, C, H, W, weight.shape[2], weight.shape[3], padding[0], padding[1],
                                           dilation[0], dilation[1], _ptr(x), _ptr(x2), _ptr(weight), _ptr(bias),
                                           _ptr(out), code, _stream(x)), "unetmi_dwconv2d_fwd")
    return out


def _layernorm_f_cuda(x, gamma, beta, eps, out, c_off):
    code = dtype_code(x)
    _dev(x, "x", "layernorm_f", x.dtype), _dev(out, "out", "layernorm_f", x.dtype)
    N, C, F, T = x.shape
    if out.shape[0] != N or tuple(out.shape[2:]) != (F, T) or gamma.numel() != F or beta.numel() != F:
        raise ValueError("unetmi::layernorm_f: shape mismatch")
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_layernorm_f_fwd(N, C, F, T, out.shape[1], c_off, eps, _ptr(x), _ptr(gamma), _ptr(beta),
                                              _ptr(out), code, _stream(x)), "unetmi_layernorm_f_fwd")


def _tanh_mask_mul_cuda(y, x, eps):
    code = dtype_code(y)
    _dev(y, "y", "tanh_mask_mul", y.dtype), _dev(x, "x", "tanh_mask_mul", y.dtype)
    N, K, F, T = y.shape
    if x.shape[0] != N or x.shape[1] < K or tuple(x.shape[2:]) != (F, T):
        raise ValueError("unetmi::tanh_mask_mul: x must be [N, >=K, F, T]")
    out = torch.empty_like(y)
    with torch.cuda.device(y.device):
        check(_lib.lib.unetmi_tanh_mask_mul(N, K, x.shape[1], F, T, eps, _ptr(y), _ptr(x), _ptr(out), code, _stream(y)),
              "unetmi_tanh_mask_mul")
    return out


def _crf_apply_cuda(H, X):
    if not (H.is_cuda and X.is_cuda and H.dtype == torch.complex64 and X.dtype == torch.complex64
            and H.is_contiguous() and X.is_contiguous() and H.dim() == 7 and X.dim() == 4):
        raise ValueError("unetmi::crf_apply: H [B,N,C,PF,PT,F,T] and X [B,C,F,T] must be contiguous complex64 "
                         "device tensors")
    B, N, C, PF, PT, F, T = H.shape
    if tuple(X.shape) != (B, C, F, T):
        raise ValueError("unetmi::crf_apply: X is {}, expected {}".format(tuple(X.shape), (B, C, F, T)))
    Y = torch.empty((B, N, C, F, T), dtype=torch.complex64, device=H.device)
    with torch.cuda.device(H.device):
        check(_lib.lib.unetmi_crf_apply(B, N, C, PF, PT, F, T, torch.view_as_real(H).data_ptr(),
                                        torch.view_as_real(X).data_ptr(), torch.view_as_real(Y).data_ptr(),
                                        _stream(H)), "unetmi_crf_apply")
    return Y


def _resblock_layer_cuda(x, x2, dw_weight, dw_bias, pw_weight, pw_bias, alpha, scale, shift, dilation, residual):
    code = dtype_code(x)
    _dev(x, "x", "resblock_layer", x.dtype)
    N, C, H, W = x.shape
    for t, name, numel in ((dw_weight, "dw_weight", C * 9), (pw_weight, "pw_weight", C * C), (alpha, "alpha", 1),
                           (scale, "scale", C), (shift, "shift", C)):
        _dev(t, name, "resblock_layer")
        if t.numel() != numel:
            raise ValueError("unetmi::resblock_layer: {} has {} elements, expected {}".format(name, t.numel(), numel))
    if x2 is not None:
        _dev(x2, "x2", "resblock_layer", x.dtype)
        if x2.shape != x.shape:
            raise ValueError("unetmi::resblock_layer: x2 shape differs from x")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_resblock_layer_fwd(N, C, H, W, dilation, _ptr(x), _ptr(x2), _ptr(dw_weight), _ptr(dw_bias),
                                                 _ptr(pw_weight), _ptr(pw_bias), _ptr(alpha), _ptr(scale), _ptr(shift),
                                                 int(residual), _ptr(out), code, _stream(x)), "unetmi_resblock_layer_fwd")
    return out


def _pwconv64_cuda(x, x2, weight, bias, scale, shift, slope, act_first, residual):
    code = dtype_code(x)
    _dev(x, "x", "pwconv64", x.dtype)
    N, C, H, W = x.shape
    if C != 64 or weight.numel() != 64 * 64:
        raise ValueError("unetmi::pwconv64: built for 64 -> 64 channels")
    for t, name in ((weight, "weight"), (scale, "scale"), (shift, "shift")):
        _dev(t, name, "pwconv64")
    if x2 is not None:
        _dev(x2, "x2", "pwconv64", x.dtype)
        if x2.shape != x.shape:
            raise ValueError("unetmi::pwconv64: x2 shape differs from x")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_pwconv64_fwd(N, H, W, _ptr(x), _ptr(x2), _ptr(weight), _ptr(bias), _ptr(scale), _ptr(shift),
                                           _ptr(slope), int(act_first), int(residual), _ptr(out), code, _stream(x)),
              "unetmi_pwconv64_fwd")
    return out


_LIBRARY.impl("resblock_layer", _resblock_layer_cuda, "CUDA")
def _glu_cuda(z, scale, shift, act, act_param, gate):
    code = dtype_code(z)
    _dev(z, "z", "glu", z.dtype)
    N, K2 = z.shape[0], z.shape[1]
    if K2 % 2:
        raise ValueError("unetmi::glu: z needs an even number of channels (features, then mask logits)")
    K = K2 // 2
    HW = z.numel() // max(N * K2, 1)
    for t, name, numel in ((scale, "scale", K), (shift, "shift", K), (act_param, "act_param", 1), (gate, "gate", N * K)):
        if t is not None:
            _dev(t, name, "glu")
            if t.numel() != numel:
                raise ValueError("unetmi::glu: {} has {} elements, expected {}".format(name, t.numel(), numel))
    out = torch.empty((N, K) + tuple(z.shape[2:]), dtype=z.dtype, device=z.device)
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_glu_fwd(N, K, HW, _ptr(z), _ptr(scale), _ptr(shift), act, _ptr(act_param), _ptr(gate),
                                      _ptr(out), code, _stream(z)), "unetmi_glu_fwd")
    return out


_LIBRARY.impl("glu", _glu_cuda, "CUDA")
_LIBRARY.impl("pwconv64", _pwconv64_cuda, "CUDA")
_LIBRARY.impl("dwconv2d_fwd", _dwconv2d_fwd_cuda, "CUDA")
_LIBRARY.impl("layernorm_f", _layernorm_f_cuda, "CUDA")
_LIBRARY.impl("tanh_mask_mul", _tanh_mask_mul_cuda, "CUDA")
_LIBRARY.impl("crf_apply", _crf_apply_cuda, "CUDA")


@torch.library.register_fake("unetmi::dwconv2d_fwd")
def _(x, x2, weight, bias, padding, dilation):
    return torch.empty_like(x)


@torch.library.register_fake("unetmi::glu")
def _(z, scale, shift, act, act_param, gate):
    return z.new_empty((z.shape[0], z.shape[1] // 2) + tuple(z.shape[2:]))


@torch.library.register_fake("unetmi::pwconv64")
def _(x, x2, weight, bias, scale, shift, slope, act_first, residual):
    return torch.empty_like(x)


@torch.library.register_fake("unetmi::resblock_layer")
def _(x, x2, dw_weight, dw_bias, pw_weight, pw_bias, alpha, scale, shift, dilation, residual):
    return torch.empty_like(x)


@torch.library.register_fake("unetmi::tanh_mask_mul")
def _(y, x, eps):
    return torch.empty_like(y)


@torch.library.register_fake("unetmi::crf_apply")
def _(H, X):
    return H.new_empty((H.shape[0], H.shape[1], H.shape[2], H.shape[5], H.shape[6]))


dwconv2d_fwd = torch.ops.unetmi.dwconv2d_fwd
layernorm_f = torch.ops.unetmi.layernorm_f
tanh_mask_mul = torch.ops.unetmi.tanh_mask_mul
crf_apply = torch.ops.unetmi.crf_apply
resblock_layer = torch.ops.unetmi.resblock_layer
pwconv64 = torch.ops.unetmi.pwconv64
glu = torch.ops.unetmi.glu
