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
    transposed, in_mode, act, act_first = flags[:4]     # optional fifth entry: row tiles per wave (0 | 1 | 2 | 4 | 5 | 6)
    d = ConvDesc()
    d.N, d.C0, d.C1, d.H, d.W, d.K = N, C0, C1, H, W, K
    d.OH = conv_out_size(H, kh, sh, ph, dh, oph, transposed)
    d.OW = conv_out_size(W, kw, sw, pw, dw, opw, transposed)
    d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, d.dh, d.dw, d.oph, d.opw = kh, kw, sh, sw, ph, pw, dh, dw, oph, opw
    d.transposed, d.in_mode, d.act, d.act_first = transposed, in_mode, act, act_first
    d.act_param_per_channel, d.has_gate, d.has_residual = act_pc, has_gate, has_residual
    d.reserved[0] = dtype
    d.reserved[1] = flags[4] if len(flags) > 4 else 0
    return d


def pack_conv_weights(weight, C0, C1, geom, flags):
    """Host-side repack of a torch-layout weight into the kernel's image
    (unetmi_conv2d_pack_weights).  Returns a CPU float32 tensor."""
    flags = list(flags)[:4]
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


def conv_kernel_name(desc):
    """The kernel instantiation unetmi_conv2d_fwd launches for ``desc`` (unetmi_conv2d_kernel_name), as a profiler
    prints it without namespace, storage-type parameter and argument list."""
    buf = ctypes.create_string_buffer(128)
    check(_lib.lib.unetmi_conv2d_kernel_name(ctypes.byref(desc), buf, len(buf)), "unetmi_conv2d_kernel_name")
    return buf.value.decode()


def conv_b16_kernel_name(desc, head_channels=0):
    """The kernel instantiation the bf16-operand entry points launch for ``desc`` (unetmi_conv2d_b16_kernel_name)."""
    buf = ctypes.create_string_buffer(128)
    check(_lib.lib.unetmi_conv2d_b16_kernel_name(ctypes.byref(desc), int(head_channels), buf, len(buf)),
          "unetmi_conv2d_b16_kernel_name")
    return buf.value.decode()


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

_LIBRARY.define("conv2d_fwd_stats(Tensor in0, Tensor? in1, Tensor wpacked, Tensor bias, Tensor scale, Tensor shift, float eps, "
                "int[] geom, int[] flags) -> (Tensor, Tensor, Tensor)")


def conv2d_fwd_stats_supported(in0, in1, geom, flags):
    """The layer's kernel has the statistics epilogue (unetmi_conv2d_fwd_stats_workspace_bytes != 0)."""
    if in0.dtype != torch.float32:
        return False
    d = make_desc(in0.shape[0], in0.shape[1], 0 if in1 is None else in1.shape[1], in0.shape[2], in0.shape[3], geom, flags,
                  0, 0, 0, dtype_code(in0))
    return in0.shape[0] > 0 and _lib.lib.unetmi_conv2d_fwd_stats_workspace_bytes(ctypes.byref(d)) != 0


def _conv2d_fwd_stats_cuda(in0, in1, wpacked, bias, scale, shift, eps, geom, flags):
    """(z, mean, invstd): the convolution (no activation) and the batch statistics of its output, the per-channel
    sums taken in the convolution's epilogue.  flags = [transposed, in_mode, ACT_NONE, 0]."""
    _need(in0, "in0")
    N, C0, H, W = in0.shape
    C1 = 0
    if in1 is not None:
        _need(in1, "in1")
        if in1.shape[0] != N or tuple(in1.shape[2:]) != (H, W):
            raise ValueError("unetmi::conv2d_fwd_stats: in1 {} does not match in0 {}".format(tuple(in1.shape), tuple(in0.shape)))
        C1 = in1.shape[1]
    K = geom[0]
    d = make_desc(N, C0, C1, H, W, geom, flags, 0, 0, 0, dtype_code(in0))
    for t, name in ((bias, "bias"), (scale, "scale"), (shift, "shift")):
        _need(t, name, (K,))
    _need(wpacked, "wpacked")
    expect = _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d))
    if expect == 0:
        raise _lib.UnetmiError("unetmi::conv2d_fwd_stats: " + _lib.last_error())
    if wpacked.numel() * 4 != expect:
        raise ValueError("unetmi::conv2d_fwd_stats: packed weight image has {} bytes, expected {}".format(wpacked.numel() * 4, expect))
    need = _lib.lib.unetmi_conv2d_fwd_stats_workspace_bytes(ctypes.byref(d))
    if need == 0:
        raise _lib.UnetmiError("unetmi::conv2d_fwd_stats: " + _lib.last_error(), _lib.E_UNSUPPORTED)
    out = torch.empty((N, K, d.OH, d.OW), dtype=torch.float32, device=in0.device)
    mean = torch.empty((K,), dtype=torch.float32, device=in0.device)
    invstd = torch.empty((K,), dtype=torch.float32, device=in0.device)
    ws = workspace(in0.device, need)
    with torch.cuda.device(in0.device):
        check(_lib.lib.unetmi_conv2d_fwd_stats(ctypes.byref(d), _ptr(in0), _ptr(in1), _ptr(wpacked), _ptr(bias), _ptr(scale),
                                               _ptr(shift), _ptr(out), float(eps), _ptr(mean), _ptr(invstd), _ptr(ws), need,
                                               torch.cuda.current_stream().cuda_stream), "unetmi_conv2d_fwd_stats")
    return out, mean, invstd


_LIBRARY.impl("conv2d_fwd_stats", _conv2d_fwd_stats_cuda, "CUDA")


@torch.library.register_fake("unetmi::conv2d_fwd_stats")
def _(in0, in1, wpacked, bias, scale, shift, eps, geom, flags):
    z = _conv2d_fwd_fake(in0, in1, wpacked, bias, scale, shift, None, None, None, geom, flags)
    return z, z.new_empty((geom[0],)), z.new_empty((geom[0],))


conv2d_fwd_stats = torch.ops.unetmi.conv2d_fwd_stats


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
_LIBRARY.define("instnorm_act(Tensor z, Tensor? gamma, Tensor? beta, float eps, int act, Tensor? act_param, Tensor? gate) -> Tensor")
_LIBRARY.define("layernorm_w_act(Tensor z, Tensor? gamma, Tensor? beta, float eps, int act, Tensor? act_param, Tensor? gate) -> Tensor")
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


def _instnorm_act_cuda(z, gamma, beta, eps, act, act_param, gate):
    code = dtype_code(z)
    _dev(z, "z", "instnorm_act", z.dtype)
    N, C = z.shape[0], z.shape[1]
    HW = z.numel() // max(N * C, 1)
    for t, name, numel in ((gamma, "gamma", C), (beta, "beta", C), (gate, "gate", N * C)):
        if t is not None:
            _dev(t, name, "instnorm_act")
            if t.numel() != numel:
                raise ValueError("unetmi::instnorm_act: {} has {} elements, expected {}".format(name, t.numel(), numel))
    act_pc = 0
    if act_param is not None:
        _dev(act_param, "act_param", "instnorm_act")
        if act_param.numel() not in (1, C):
            raise ValueError("unetmi::instnorm_act: act_param needs 1 or {} elements".format(C))
        act_pc = int(act_param.numel() > 1)
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_instnorm_act_fwd(N, C, HW, _ptr(z), _ptr(gamma), _ptr(beta), eps, act, _ptr(act_param), act_pc,
                                               _ptr(gate), _ptr(out), code, _stream(z)), "unetmi_instnorm_act_fwd")
    return out


def _layernorm_w_act_cuda(z, gamma, beta, eps, act, act_param, gate):
    """act(LayerNorm over the last axis of z [N,C,H,W], gamma / beta [W]) * gate[n,c] as one launch."""
    code = dtype_code(z)
    _dev(z, "z", "layernorm_w_act", z.dtype)
    if z.dim() != 4:
        raise ValueError("unetmi::layernorm_w_act: z must be [N,C,H,W]")
    N, C, H, W = z.shape
    for t, name, numel in ((gamma, "gamma", W), (beta, "beta", W), (gate, "gate", N * C)):
        if t is not None:
            _dev(t, name, "layernorm_w_act")
            if t.numel() != numel:
                raise ValueError("unetmi::layernorm_w_act: {} has {} elements, expected {}".format(name, t.numel(), numel))
    act_pc = 0
    if act_param is not None:
        _dev(act_param, "act_param", "layernorm_w_act")
        if act_param.numel() not in (1, C):
            raise ValueError("unetmi::layernorm_w_act: act_param needs 1 or {} elements".format(C))
        act_pc = int(act_param.numel() > 1)
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_layernorm_w_act_fwd(N, C, H, W, _ptr(z), _ptr(gamma), _ptr(beta), eps, act, _ptr(act_param), act_pc,
                                                  _ptr(gate), _ptr(out), code, _stream(z)), "unetmi_layernorm_w_act_fwd")
    return out


_LIBRARY.impl("instnorm_act", _instnorm_act_cuda, "CUDA")
_LIBRARY.impl("layernorm_w_act", _layernorm_w_act_cuda, "CUDA")
_LIBRARY.impl("glu", _glu_cuda, "CUDA")
_LIBRARY.impl("pwconv64", _pwconv64_cuda, "CUDA")
_LIBRARY.impl("dwconv2d_fwd", _dwconv2d_fwd_cuda, "CUDA")
_LIBRARY.impl("layernorm_f", _layernorm_f_cuda, "CUDA")
_LIBRARY.impl("tanh_mask_mul", _tanh_mask_mul_cuda, "CUDA")
_LIBRARY.impl("crf_apply", _crf_apply_cuda, "CUDA")


@torch.library.register_fake("unetmi::dwconv2d_fwd")
def _(x, x2, weight, bias, padding, dilation):
    return torch.empty_like(x)


@torch.library.register_fake("unetmi::instnorm_act")
def _(z, gamma, beta, eps, act, act_param, gate):
    return torch.empty_like(z)


@torch.library.register_fake("unetmi::layernorm_w_act")
def _(z, gamma, beta, eps, act, act_param, gate):
    return torch.empty_like(z)


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
instnorm_act = torch.ops.unetmi.instnorm_act
layernorm_w_act = torch.ops.unetmi.layernorm_w_act


# ---------------------------------------------------------------------------------------
# training entry points (csrc/bn_train.hip, csrc/wgrad.hip), the recurrence kernels (csrc/lstm*.hip) and the
# bf16-operand convolution on the blocked layout (csrc/conv_b16.hip)
# ---------------------------------------------------------------------------------------
_LIBRARY.define("bn_train_stats(Tensor z, float eps, float momentum, Tensor(a!) running_mean, Tensor(b!) running_var)"
                " -> (Tensor, Tensor)")
_LIBRARY.define("bn_batch_stats(Tensor z, float eps) -> (Tensor, Tensor)")
_LIBRARY.define("bn_running_update(Tensor(a!) running_mean, Tensor(b!) running_var, Tensor mean, Tensor invstd, int count, "
                "float eps, float momentum) -> ()")
_LIBRARY.define("bn_act_fwd(Tensor z, Tensor mean, Tensor invstd, Tensor gamma, Tensor beta, int act, Tensor? act_param)"
                " -> Tensor")
_LIBRARY.define("bn_act_bwd(Tensor dy, Tensor z, Tensor mean, Tensor invstd, Tensor gamma, Tensor beta, int act, "
                "Tensor? act_param, bool batch_stats) -> (Tensor, Tensor, Tensor, Tensor)")
_LIBRARY.define("conv2d_wgrad(Tensor in0, Tensor? in1, Tensor dout, int[] geom, int[] flags) -> Tensor")
_LIBRARY.define("adam_step(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, "
                "float beta2, float eps, float weight_decay, float grad_scale, int step) -> ()")
_LIBRARY.define("gather_f32(Tensor src, Tensor map) -> Tensor")
_LIBRARY.define("dwconv2d_wgrad(Tensor x, Tensor? x2, Tensor dout, int[] kernel, int[] padding, int[] dilation) -> (Tensor, Tensor)")
_LIBRARY.define("lstm_recurrence(Tensor wrec, Tensor gx, Tensor(a!) hfull, bool xcd, bool channel_major, bool f16=False) -> Tensor")
_LIBRARY.define("lstm_recurrence_train(Tensor wrec, Tensor(a!) gx, Tensor(b!) hfull, Tensor(c!) cell) -> Tensor")
_LIBRARY.define("lstm_bptt(Tensor wrec, Tensor(a!) gates, Tensor cell, Tensor dhfull) -> Tensor")
_LIBRARY.define("gru_recurrence(Tensor whh, Tensor bhh, Tensor gx, Tensor(a!) hfull, bool channel_major) -> Tensor")
_LIBRARY.define("gru_recurrence_train(Tensor whh, Tensor bhh, Tensor gx, Tensor(a!) hfull, Tensor(b!) gates) -> Tensor")
_LIBRARY.define("gru_bptt(Tensor whh, Tensor(a!) gates, Tensor hfull, Tensor dy) -> (Tensor, Tensor)")
_LIBRARY.define("gru_gate(Tensor gx, int step, int direction, Tensor? gh, Tensor bhh, Tensor(a!) y, int prev_step) -> ()")
_LIBRARY.define("blocked_from_nchw(Tensor x) -> Tensor")
_LIBRARY.define("nchw_from_blocked(Tensor data, int C, bool f32) -> Tensor")
_LIBRARY.define("conv2d_b16_fwd(Tensor x, int C0, Tensor? skip, int C1, Tensor wpacked, Tensor bias, Tensor scale, "
                "Tensor shift, Tensor? act_param, Tensor? gate, int[] geom, int[] flags) -> Tensor")
_LIBRARY.define("blocked_from_nchw_sum(Tensor x, Tensor? x2, bool f16=False) -> Tensor")
_LIBRARY.define("conv2d_b16_nchw_fwd(Tensor x, int C0, Tensor? skip, int C1, Tensor wpacked, Tensor bias, Tensor scale, "
                "Tensor shift, Tensor? act_param, Tensor? gate, Tensor? residual, int[] geom, int[] flags) -> Tensor")
_LIBRARY.define("conv2d_b16_head_fwd(Tensor x, int C0, Tensor? skip, int C1, Tensor wpacked, Tensor bias, Tensor scale, "
                "Tensor shift, Tensor? act_param, Tensor? gate, Tensor head_w, Tensor? head_b, int[] geom, int[] flags) -> Tensor")
# conv2d_wgrad flags = [transposed, in_mode]; conv2d_b16_fwd flags = [transposed, act, act_first, halves, nchw_out(, out_f16)]
# (the operands' 16-bit format, bfloat16 or fp16, is the dtype of x; out_f16 asks for an fp16 blocked output);
# conv2d_b16_head_fwd flags = [transposed, act, act_first, halves, head_act, interleaved];
# conv2d_b16_nchw_fwd flags = [transposed, act, act_first, halves]

_ws_cache = {}


def workspace(device, nbytes):
    """One growing scratch buffer per (device, stream): its contents never outlive a launch sequence, and work on
    different streams never shares one."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _nc_hw(z):
    N, C = z.shape[0], z.shape[1]
    return N, C, z.numel() // max(N * C, 1)


def _bn_train_stats_cuda(z, eps, momentum, running_mean, running_var):
    _dev(z, "z", "bn_train_stats")
    N, C, HW = _nc_hw(z)
    for t, name in ((running_mean, "running_mean"), (running_var, "running_var")):
        _dev(t, name, "bn_train_stats")
        if t.numel() != C:
            raise ValueError("unetmi::bn_train_stats: {} needs {} elements".format(name, C))
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    invstd = torch.empty_like(mean)
    ws = workspace(z.device, _lib.lib.unetmi_bn_workspace_bytes(N, C, HW))
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_bn_train_stats(N, C, HW, _ptr(z), eps, momentum, _ptr(running_mean), _ptr(running_var),
                                             _ptr(mean), _ptr(invstd), _ptr(ws), ws.numel(), _stream(z)),
              "unetmi_bn_train_stats")
    return mean, invstd


def _bn_batch_stats_cuda(z, eps):
    """(mean, invstd) over all but the channel axis, without touching any running statistic."""
    _dev(z, "z", "bn_batch_stats")
    N, C, HW = _nc_hw(z)
    mean = torch.empty(C, dtype=torch.float32, device=z.device)
    invstd = torch.empty_like(mean)
    ws = workspace(z.device, _lib.lib.unetmi_bn_workspace_bytes(N, C, HW))
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_bn_train_stats(N, C, HW, _ptr(z), eps, 0.0, None, None, _ptr(mean), _ptr(invstd), _ptr(ws),
                                             ws.numel(), _stream(z)), "unetmi_bn_train_stats")
    return mean, invstd


def _bn_running_update_cuda(running_mean, running_var, mean, invstd, count, eps, momentum):
    C = mean.numel()
    for t, name in ((running_mean, "running_mean"), (running_var, "running_var"), (mean, "mean"), (invstd, "invstd")):
        _dev(t, name, "bn_running_update")
        if t.numel() != C:
            raise ValueError("unetmi::bn_running_update: {} needs {} elements".format(name, C))
    with torch.cuda.device(mean.device):
        check(_lib.lib.unetmi_bn_running_update(C, count, eps, momentum, _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                                _ptr(running_var), _stream(mean)), "unetmi_bn_running_update")


def _bn_act_fwd_cuda(z, mean, invstd, gamma, beta, act, act_param):
    _dev(z, "z", "bn_act_fwd")
    N, C, HW = _nc_hw(z)
    for t, name in ((mean, "mean"), (invstd, "invstd"), (gamma, "gamma"), (beta, "beta")):
        _dev(t, name, "bn_act_fwd")
        if t.numel() != C:
            raise ValueError("unetmi::bn_act_fwd: {} needs {} elements".format(name, C))
    out = torch.empty_like(z)
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_bn_act_fwd(N, C, HW, _ptr(z), _ptr(mean), _ptr(invstd), _ptr(gamma), _ptr(beta), act,
                                         _ptr(act_param), _ptr(out), _stream(z)), "unetmi_bn_act_fwd")
    return out


def _bn_act_bwd_cuda(dy, z, mean, invstd, gamma, beta, act, act_param, batch_stats):
    _dev(z, "z", "bn_act_bwd")
    N, C, HW = _nc_hw(z)
    if dy.shape != z.shape or dy.dtype != torch.float32 or not dy.is_cuda:
        raise ValueError("unetmi::bn_act_bwd: dy must match z")
    # dy may be a channel slice of a wider contiguous tensor: same inner strides, larger batch stride
    inner = tuple(dy.stride()[1:]) == tuple(z.stride()[1:])
    if not inner or (N > 1 and dy.stride(0) < C * HW):
        dy = dy.contiguous()
    batch_stride = dy.stride(0) if N > 1 else C * HW
    dz = torch.empty_like(z)
    dgamma, dbeta, dbias = (torch.empty(C, dtype=torch.float32, device=z.device) for _ in range(3))
    ws = workspace(z.device, _lib.lib.unetmi_bn_workspace_bytes(N, C, HW))
    with torch.cuda.device(z.device):
        check(_lib.lib.unetmi_bn_act_bwd(N, C, HW, _ptr(dy), batch_stride, _ptr(z), _ptr(mean), _ptr(invstd),
                                         _ptr(gamma), _ptr(beta), act, _ptr(act_param), int(batch_stats), _ptr(dz),
                                         _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(ws), ws.numel(), _stream(z)),
              "unetmi_bn_act_bwd")
    return dz, dgamma, dbeta, dbias


def _wgrad_shape(C, geom, transposed):
    return (C, geom[0], geom[1], geom[2]) if transposed else (geom[0], C, geom[1], geom[2])


def _conv2d_wgrad_cuda(in0, in1, dout, geom, flags):
    _dev(in0, "in0", "conv2d_wgrad"), _dev(dout, "dout", "conv2d_wgrad")
    N, C0, H, W = in0.shape
    C1 = 0
    if in1 is not None:
        _dev(in1, "in1", "conv2d_wgrad")
        C1 = in1.shape[1]
    desc = make_desc(N, C0, C1, H, W, geom, [flags[0], flags[1], 0, 0])
    if tuple(dout.shape) != (N, geom[0], desc.OH, desc.OW):
        raise ValueError("unetmi::conv2d_wgrad: dout is {}, expected {}".format(tuple(dout.shape), (N, geom[0], desc.OH, desc.OW)))
    dw = torch.empty(_wgrad_shape(C0 + C1, geom, flags[0]), dtype=torch.float32, device=in0.device)
    nbytes = _lib.lib.unetmi_conv2d_wgrad_workspace_bytes(ctypes.byref(desc))
    if nbytes == 0:
        raise _lib.UnetmiError("unetmi_conv2d_wgrad: " + _lib.last_error(), _lib.E_UNSUPPORTED)
    ws = workspace(in0.device, nbytes)
    with torch.cuda.device(in0.device):
        check(_lib.lib.unetmi_conv2d_wgrad(ctypes.byref(desc), _ptr(in0), _ptr(in1), _ptr(dout), _ptr(dw), _ptr(ws),
                                           ws.numel(), _stream(in0)), "unetmi_conv2d_wgrad")
    return dw


def _adam_step_cuda(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    for t in (param, grad, exp_avg, exp_avg_sq):
        _dev(t, "bucket", "adam_step")
        if t.numel() != param.numel():
            raise ValueError("unetmi::adam_step: bucket sizes differ")
    with torch.cuda.device(param.device):
        check(_lib.lib.unetmi_adam_step(param.numel(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), lr,
                                        beta1, beta2, eps, weight_decay, grad_scale, step, _stream(param)),
              "unetmi_adam_step")


def _gather_f32_cuda(src, map):
    """out[i] = src.flatten()[map[i]], 0 where map[i] < 0 (int32 map): the device-side weight re-pack."""
    _dev(src, "src", "gather_f32")
    _dev(map, "map", "gather_f32", torch.int32)
    out = torch.empty(map.numel(), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        check(_lib.lib.unetmi_gather_f32(map.numel(), _ptr(src), _ptr(map), _ptr(out), _stream(src)), "unetmi_gather_f32")
    return out


# flag / status words of the persistent recurrence kernels, one buffer per (device, stream, kind, H): a launch
# clears its flags with a memset on ITS stream, so two streams must not share the words of a kernel still spinning
_lstm_scratch = {}


def _lstm_recurrence_cuda(wrec, gx, hfull, xcd, channel_major, f16=False):
    """One layer's T dependent steps with the folded recurrent matrix wrec [4H, H].  Step-major: gx [T, 4H, BP]
    (input contributions + bias), hfull [T+1, HP, BP], HP = H rounded up to 16 (slot 0 = initial state, slots 1..
    are filled; rows H..HP-1 must be zero).  Channel-major
    (XCD kernel only): gx [4H, T+1, BP] with slot t+1 = step t, hfull [H, T+1, BP].  f16 (XCD kernel only): the
    recurrent product takes fp16 operands (fp32 accumulation and state).  Returns the kernel's status
    word as a device scalar (non-zero: a wait gave up and the remaining steps are NaN)."""
    for t, name in ((wrec, "wrec"), (gx, "gx"), (hfull, "hfull")):
        _dev(t, name, "lstm_recurrence")
    if channel_major:
        G, Tp, BP = gx.shape
        T, H = Tp - 1, G // 4
        ok = tuple(hfull.shape) == (H, Tp, BP) and xcd
    else:
        T, G, BP = gx.shape
        H = G // 4
        ok = tuple(hfull.shape) == (T + 1, (H + 15) // 16 * 16, BP)       # a slot's rows are padded to a multiple of 16
    if f16 and not xcd:
        raise ValueError("unetmi::lstm_recurrence: fp16 operands are built into the XCD-local kernel only")
    if not ok or G != 4 * H or tuple(wrec.shape) != (G, H):
        raise ValueError("unetmi::lstm_recurrence: wrec [4H,H] with gx [T,4H,BP], hfull [T+1,H,BP], or (XCD kernel) "
                         "gx [4H,T+1,BP], hfull [H,T+1,BP] expected")
    with torch.cuda.device(gx.device):
        stream = _stream(gx)
        key = (gx.device, stream, bool(xcd), H)
        if xcd:
            nbytes = _lib.lib.unetmi_lstm_recurrence_xcd_workspace_bytes()
            ws = _lstm_scratch.get(key)
            if ws is None:
                ws = _lstm_scratch[key] = torch.empty(nbytes // 4, dtype=torch.int32, device=gx.device)
            check(_lib.lib.unetmi_lstm_recurrence_xcd(T, BP, H, _ptr(wrec), _ptr(gx), _ptr(hfull),
                                                      int(channel_major) | (2 if f16 else 0),
                                                      _ptr(ws), nbytes, stream), "unetmi_lstm_recurrence_xcd")
            return ws[8].clone()
        sync = _lstm_scratch.get(key)
        if sync is None or sync.numel() != H // 4 + 1:
            sync = _lstm_scratch[key] = torch.empty(H // 4 + 1, dtype=torch.int32, device=gx.device)
        check(_lib.lib.unetmi_lstm_recurrence(T, BP, H, _ptr(wrec), _ptr(gx), _ptr(hfull), _ptr(sync), stream),
              "unetmi_lstm_recurrence")
        return sync[-1].clone()


def _lstm_train_shapes(op, wrec, gates, others):
    for t, name in ((wrec, "wrec"), (gates, "gates")) + tuple(others):
        _dev(t, name, op)
    G, Tp, BP = gates.shape
    H = G // 4
    if G != 4 * H or tuple(wrec.shape) != (G, H) or any(tuple(t.shape) != (H, Tp, BP) for t, _ in others):
        raise ValueError("unetmi::{}: wrec [4H,H], gates [4H,T+1,BP] and [H,T+1,BP] state tensors expected".format(op))
    return Tp - 1, BP, H


def _lstm_recurrence_train_cuda(wrec, gx, hfull, cell):
    """The XCD-local forward recurrence on channel-major tensors, keeping the activated gates (written over gx) and
    the cell states for lstm_bptt.  Returns the status word."""
    T, BP, H = _lstm_train_shapes("lstm_recurrence_train", wrec, gx, ((hfull, "hfull"), (cell, "cell")))
    with torch.cuda.device(gx.device):
        stream = _stream(gx)
        key = (gx.device, stream, True, H)
        nbytes = _lib.lib.unetmi_lstm_recurrence_xcd_workspace_bytes()
        ws = _lstm_scratch.get(key)
        if ws is None:
            ws = _lstm_scratch[key] = torch.empty(nbytes // 4, dtype=torch.int32, device=gx.device)
        check(_lib.lib.unetmi_lstm_recurrence_xcd_train(T, BP, H, _ptr(wrec), _ptr(gx), _ptr(hfull), _ptr(cell), _ptr(ws),
                                                        nbytes, stream), "unetmi_lstm_recurrence_xcd_train")
        return ws[8].clone()


def _lstm_bptt_cuda(wrec, gates, cell, dhfull):
    """The reverse recurrence: gates [4H, T+1, BP] (as lstm_recurrence_train left them) become the gradients of the
    gate pre-activations in place.  Returns the status word."""
    T, BP, H = _lstm_train_shapes("lstm_bptt", wrec, gates, ((cell, "cell"), (dhfull, "dhfull")))
    with torch.cuda.device(gates.device):
        stream = _stream(gates)
        key = (gates.device, stream, "bptt", H)
        nbytes = _lib.lib.unetmi_lstm_bptt_xcd_workspace_bytes()
        ws = _lstm_scratch.get(key)
        if ws is None:
            ws = _lstm_scratch[key] = torch.empty(nbytes // 4, dtype=torch.int32, device=gates.device)
        check(_lib.lib.unetmi_lstm_bptt_xcd(T, BP, H, _ptr(wrec), _ptr(gates), _ptr(cell), _ptr(dhfull), _ptr(ws), nbytes,
                                            stream), "unetmi_lstm_bptt_xcd")
        return ws[8].clone()


def _dwconv2d_wgrad_cuda(x, x2, dout, kernel, padding, dilation):
    """(dweight [C,1,kh,kw], dbias [C]) of the depthwise 'same' convolution that dwconv2d_fwd computes."""
    _dev(x, "x", "dwconv2d_wgrad"), _dev(dout, "dout", "dwconv2d_wgrad")
    if x2 is not None:
        _dev(x2, "x2", "dwconv2d_wgrad")
    if dout.shape != x.shape or (x2 is not None and x2.shape != x.shape):
        raise ValueError("unetmi::dwconv2d_wgrad: x, x2 and dout must have one shape")
    N, C, H, W = x.shape
    kh, kw = kernel
    dweight = torch.empty((C, 1, kh, kw), dtype=torch.float32, device=x.device)
    dbias = torch.empty((C,), dtype=torch.float32, device=x.device)
    need = _lib.lib.unetmi_dwconv2d_wgrad_workspace_bytes(C, kh, kw)
    ws = workspace(x.device, need)
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_dwconv2d_wgrad(N, C, H, W, kh, kw, padding[0], padding[1], dilation[0], dilation[1], _ptr(x),
                                             _ptr(x2), _ptr(dout), _ptr(dweight), _ptr(dbias), _ptr(ws), need, _stream(x)),
              "unetmi_dwconv2d_wgrad")
    return dweight, dbias


def _gru_recurrence_cuda(whh, bhh, gx, hfull, channel_major):
    """One GRU layer's T dependent steps: gx [T, 3H, BP] (channel_major: [3H, T, BP]; input contributions + b_ih),
    whh [3H, H], bhh [3H], hfull [T+1, H, BP] with slot 0 = the initial state; fills hfull[1:].  Returns the status
    word as a device scalar."""
    for t, name in ((whh, "whh"), (bhh, "bhh"), (gx, "gx"), (hfull, "hfull")):
        _dev(t, name, "gru_recurrence")
    if channel_major:
        G, T, BP = gx.shape
    else:
        T, G, BP = gx.shape
    H = G // 3
    if G != 3 * H or tuple(whh.shape) != (G, H) or bhh.numel() != G or tuple(hfull.shape) != (T + 1, H, BP):
        raise ValueError("unetmi::gru_recurrence: whh [3H,H], bhh [3H], gx [T,3H,BP] or [3H,T,BP], hfull [T+1,H,BP] expected")
    with torch.cuda.device(gx.device):
        key = (gx.device, _stream(gx), "gru", H)
        sync = _lstm_scratch.get(key)
        if sync is None:
            sync = _lstm_scratch[key] = torch.empty(H // 4 + 1, dtype=torch.int32, device=gx.device)
        check(_lib.lib.unetmi_gru_recurrence(T, BP, H, _ptr(whh), _ptr(bhh), _ptr(gx), int(channel_major), _ptr(hfull),
                                             _ptr(sync), _stream(gx)), "unetmi_gru_recurrence")
    return sync[-1].clone()


def _gru_sync(dev, H, kind):
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, kind, H)
    sync = _lstm_scratch.get(key)
    if sync is None:
        sync = _lstm_scratch[key] = torch.empty(H // 4 + 1, dtype=torch.int32, device=dev)
    return sync


def _gru_recurrence_train_cuda(whh, bhh, gx, hfull, gates):
    """gru_recurrence on step-major tensors that also fills gates [T, 4H, BP] with r, z, q = W_hn h + b_hn and n."""
    for t, name in ((whh, "whh"), (bhh, "bhh"), (gx, "gx"), (hfull, "hfull"), (gates, "gates")):
        _dev(t, name, "gru_recurrence_train")
    T, G, BP = gx.shape
    H = G // 3
    if (G != 3 * H or tuple(whh.shape) != (G, H) or bhh.numel() != G or tuple(hfull.shape) != (T + 1, H, BP)
            or tuple(gates.shape) != (T, 4 * H, BP)):
        raise ValueError("unetmi::gru_recurrence_train: whh [3H,H], bhh [3H], gx [T,3H,BP], hfull [T+1,H,BP], gates [T,4H,BP] expected")
    with torch.cuda.device(gx.device):
        sync = _gru_sync(gx.device, H, "gru")
        check(_lib.lib.unetmi_gru_recurrence_train(T, BP, H, _ptr(whh), _ptr(bhh), _ptr(gx), _ptr(hfull), _ptr(gates),
                                                   _ptr(sync), _stream(gx)), "unetmi_gru_recurrence_train")
    return sync[-1].clone()


def _gru_bptt_cuda(whh, gates, hfull, dy):
    """The reverse recurrence: gates [T, 4H, BP] become the pre-activation gradients in place; returns (dh0 [H, BP],
    status word)."""
    for t, name in ((whh, "whh"), (gates, "gates"), (hfull, "hfull"), (dy, "dy")):
        _dev(t, name, "gru_bptt")
    T, G, BP = gates.shape
    H = G // 4
    if G != 4 * H or tuple(whh.shape) != (3 * H, H) or tuple(hfull.shape) != (T + 1, H, BP) or tuple(dy.shape) != (T, H, BP):
        raise ValueError("unetmi::gru_bptt: whh [3H,H], gates [T,4H,BP], hfull [T+1,H,BP], dy [T,H,BP] expected")
    dh0 = torch.empty((H, BP), dtype=torch.float32, device=gates.device)
    with torch.cuda.device(gates.device):
        sync = _gru_sync(gates.device, H, "gru_bptt")
        check(_lib.lib.unetmi_gru_bptt(T, BP, H, _ptr(whh), _ptr(gates), _ptr(hfull), _ptr(dy), _ptr(dh0), _ptr(sync),
                                       _stream(gates)), "unetmi_gru_bptt")
    return dh0, sync[H // 16].clone()


def _gru_gate_cuda(gx, step, direction, gh, bhh, y, prev_step):
    """One step of a (bi)directional GRU over many sequences: gx [M, L, D*3H] (input products of all steps and
    directions), gh [M, 3H] (this step's recurrent product) or None, y [M, L, D*H]; reads the previous state from
    y[:, prev_step, direction*H:...] (prev_step < 0: zero state) and writes y[:, step, direction*H:...]."""
    for t, name in ((gx, "gx"), (bhh, "bhh"), (y, "y")):
        _dev(t, name, "gru_gate")
    M, L, G = gx.shape
    H = bhh.numel() // 3
    D = G // (3 * H)
    if G != D * 3 * H or tuple(y.shape) != (M, L, D * H) or not (0 <= step < L and 0 <= direction < D and prev_step < L):
        raise ValueError("unetmi::gru_gate: gx [M,L,D*3H], bhh [3H], y [M,L,D*H] expected")
    if gh is not None:
        _dev(gh, "gh", "gru_gate")
        if tuple(gh.shape) != (M, 3 * H):
            raise ValueError("unetmi::gru_gate: gh must be [M, 3H]")
    gx_p = gx.data_ptr() + 4 * (step * G + direction * 3 * H)
    out_p = y.data_ptr() + 4 * (step * D * H + direction * H)
    prev_p = None if prev_step < 0 else y.data_ptr() + 4 * (prev_step * D * H + direction * H)
    with torch.cuda.device(gx.device):
        check(_lib.lib.unetmi_gru_gate_fwd(M, H, gx_p, L * G, _ptr(gh), _ptr(bhh), prev_p, L * D * H, out_p, L * D * H, _stream(gx)),
              "unetmi_gru_gate_fwd")


def blocks_for(C):
    return 2 * ((C + 15) // 16)


def _blocked_from_nchw_sum_cuda(x, x2, f16=False):
    if not x.is_cuda or not x.is_contiguous() or x.dim() != 4:
        raise ValueError("unetmi::blocked_from_nchw: contiguous 4-D device tensor expected")
    if x2 is not None and (x2.shape != x.shape or x2.dtype != x.dtype or x2.device != x.device or not x2.is_contiguous()):
        raise ValueError("unetmi::blocked_from_nchw_sum: x2 must match x in shape, dtype and device and be contiguous")
    N, C, H, W = x.shape
    out = torch.empty((N, blocks_for(C), H, W, 8), dtype=torch.float16 if f16 else torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        check(_lib.lib.unetmi_blocked_from_nchw_sum(N, C, H * W, _ptr(x), _ptr(x2), dtype_code(x), _ptr(out),
                                                    _lib.DTYPE_F16 if f16 else _lib.DTYPE_BF16, _stream(x)),
              "unetmi_blocked_from_nchw_sum")
    return out


def _blocked_from_nchw_cuda(x):
    return _blocked_from_nchw_sum_cuda(x, None)


def _blocked_dtype(t, name, op):
    """A blocked 16-bit tensor: bfloat16 or fp16 (its dtype names the format); True for fp16."""
    if t.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("unetmi::{}: {} must be a bfloat16 or float16 blocked tensor".format(op, name))
    _dev(t, name, op, t.dtype)
    return t.dtype == torch.float16


def _nchw_from_blocked_cuda(data, C, f32):
    src_f16 = _blocked_dtype(data, "data", "nchw_from_blocked")
    if data.dim() != 5 or data.shape[4] != 8 or data.shape[1] != blocks_for(C):
        raise ValueError("unetmi::nchw_from_blocked: data must be [N, 2*ceil(C/16), H, W, 8]")
    N, _, H, W, _ = data.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32 if f32 else torch.bfloat16, device=data.device)
    with torch.cuda.device(data.device):
        check(_lib.lib.unetmi_nchw_from_blocked(N, C, H * W, _ptr(data), _lib.DTYPE_F16 if src_f16 else _lib.DTYPE_BF16, _ptr(out),
                                                _lib.DTYPE_F32 if f32 else _lib.DTYPE_BF16, _stream(data)),
              "unetmi_nchw_from_blocked")
    return out


def b16_desc(N, C0, C1, H, W, geom, transposed, act, act_first, act_pc, has_gate, halves, nchw_out, in_f16=False,
             out_f16=False):
    """Descriptor of a 16-bit-operand layer; in_f16: inputs and weight image are fp16 (else bfloat16), out_f16: the
    blocked output is stored as fp16."""
    d = make_desc(N, C0, C1, H, W, geom, [transposed, _lib.IN_CONCAT if C1 else _lib.IN_SINGLE, act, act_first],
                  act_pc, has_gate, 0, _lib.DTYPE_BF16)
    d.reserved[1] = halves
    d.reserved[2] = ((_lib.B16_OUT_NCHW_F32 if nchw_out else 0) | (_lib.B16_IN_F16 if in_f16 else 0)
                     | (_lib.B16_OUT_F16 if out_f16 else 0))
    return d


def _conv2d_b16_run(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first,
                    halves, nchw_out, head, residual=None, out_f16=False):
    """Checks shared by the two bf16-operand entry points, then the launch.  head: None, or (head_w [P,K], head_b
    [P] | None, head_act, interleaved) for the decoder + 1x1 head launch."""
    what = "conv2d_b16_fwd" if head is None else "conv2d_b16_head_fwd"
    in_f16 = _blocked_dtype(x, "x", what)          # the operands' format: bfloat16 or fp16, both inputs alike
    if x.dim() != 5 or x.shape[4] != 8 or x.shape[1] != blocks_for(C0):
        raise ValueError("unetmi::{}: x must be [N, 2*ceil(C0/16), H, W, 8]".format(what))
    N, _, H, W, _ = x.shape
    if skip is not None:
        _dev(skip, "skip", what, x.dtype)
        if skip.dim() != 5 or skip.shape[0] != N or tuple(skip.shape[2:4]) != (H, W) or skip.shape[1] != blocks_for(C1):
            raise ValueError("unetmi::{}: skip {} does not match x {}".format(what, tuple(skip.shape), tuple(x.shape)))
    elif C1:
        raise ValueError("unetmi::{}: C1 without a skip tensor".format(what))
    K = geom[0]
    act_pc = 1 if (act_param is not None and act_param.numel() > 1) else 0
    d = b16_desc(N, C0, C1, H, W, geom, transposed, act, act_first, act_pc, int(gate is not None), halves, nchw_out,
                 in_f16, out_f16)
    expect = _lib.lib.unetmi_conv2d_b16_packed_weight_bytes(ctypes.byref(d))
    if expect == 0:
        raise _lib.UnetmiError("conv2d_b16: " + _lib.last_error(), _lib.E_BAD_DESC)
    if not wpacked.is_cuda or wpacked.numel() * wpacked.element_size() != expect:
        raise ValueError("unetmi::{}: packed weight image has {} bytes, expected {} on the device".format(
            what, wpacked.numel() * wpacked.element_size(), expect))
    for t, name in ((bias, "bias"), (scale, "scale"), (shift, "shift")):
        _dev(t, name, what)
        if tuple(t.shape) != (K,):
            raise ValueError("unetmi::{}: {} must be [K]".format(what, name))
    if act_param is not None:
        _dev(act_param, "act_param", what)
    if gate is not None:
        _dev(gate, "gate", what)
        if tuple(gate.shape) != (N, K):
            raise ValueError("unetmi::{}: gate must be [N,K]".format(what))
    if head is not None:
        head_w, head_b, head_act, interleaved = head
        _dev(head_w, "head_w", what)
        P = head_w.shape[0]
        if head_w.dim() != 2 or head_w.shape[1] != K or not 1 <= P <= 4:
            raise ValueError("unetmi::conv2d_b16_head_fwd: head_w must be [P, K] with 1 <= P <= 4")
        if head_b is not None:
            _dev(head_b, "head_b", what)
            if tuple(head_b.shape) != (P,):
                raise ValueError("unetmi::conv2d_b16_head_fwd: head_b must be [P]")
        out = torch.empty((N, d.OH, d.OW, P) if interleaved else (N, P, d.OH, d.OW), dtype=torch.float32, device=x.device)
    elif nchw_out:
        out = torch.empty((N, K, d.OH, d.OW), dtype=torch.float32, device=x.device)
        if residual is not None:
            _dev(residual, "residual", what)
            if tuple(residual.shape) != tuple(out.shape) or not residual.is_contiguous():
                raise ValueError("unetmi::{}: residual must be a contiguous fp32 [N,K,OH,OW] tensor".format(what))
    else:
        out = torch.empty((N, blocks_for(K), d.OH, d.OW, 8), dtype=torch.float16 if out_f16 else torch.bfloat16,
                          device=x.device)
    with torch.cuda.device(x.device):
        obs = _launch_observer
        if obs is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if head is None and residual is not None:
            check(_lib.lib.unetmi_conv2d_b16_nchw_fwd(ctypes.byref(d), _ptr(x), _ptr(skip), _ptr(wpacked), _ptr(bias),
                                                      _ptr(scale), _ptr(shift), _ptr(act_param), _ptr(gate),
                                                      _ptr(residual), _ptr(out), _stream(x)), "unetmi_conv2d_b16_nchw_fwd")
        elif head is None:
            check(_lib.lib.unetmi_conv2d_b16_fwd(ctypes.byref(d), _ptr(x), _ptr(skip), _ptr(wpacked), _ptr(bias),
                                                 _ptr(scale), _ptr(shift), _ptr(act_param), _ptr(gate), _ptr(out),
                                                 _stream(x)), "unetmi_conv2d_b16_fwd")
        else:
            check(_lib.lib.unetmi_conv2d_b16_head_fwd(ctypes.byref(d), _ptr(x), _ptr(skip), _ptr(wpacked), _ptr(bias),
                                                      _ptr(scale), _ptr(shift), _ptr(act_param), _ptr(gate), _ptr(head_w),
                                                      _ptr(head_b), P, int(head_act), int(interleaved), _ptr(out),
                                                      _stream(x)), "unetmi_conv2d_b16_head_fwd")
        if obs is not None:
            ev1.record()
            obs(what, d, ev0, ev1)
    return out


def _conv2d_b16_fwd_cuda(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, flags):
    transposed, act, act_first, halves, nchw_out = flags[:5]
    out_f16 = bool(flags[5]) if len(flags) > 5 else False
    return _conv2d_b16_run(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first,
                           halves, nchw_out, None, out_f16=out_f16 and not nchw_out)


def _conv2d_b16_nchw_fwd_cuda(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, residual, geom, flags):
    transposed, act, act_first, halves = flags
    return _conv2d_b16_run(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first,
                           halves, 1, None, residual)


def _conv2d_b16_head_fwd_cuda(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, head_w, head_b, geom, flags):
    transposed, act, act_first, halves, head_act, interleaved = flags
    return _conv2d_b16_run(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, transposed, act, act_first,
                           halves, 0, (head_w, head_b, head_act, interleaved))


for _name, _fn in (("bn_train_stats", _bn_train_stats_cuda), ("bn_batch_stats", _bn_batch_stats_cuda),
                   ("bn_running_update", _bn_running_update_cuda), ("bn_act_fwd", _bn_act_fwd_cuda),
                   ("bn_act_bwd", _bn_act_bwd_cuda), ("conv2d_wgrad", _conv2d_wgrad_cuda),
                   ("adam_step", _adam_step_cuda), ("gather_f32", _gather_f32_cuda), ("lstm_recurrence", _lstm_recurrence_cuda),
                   ("lstm_recurrence_train", _lstm_recurrence_train_cuda), ("lstm_bptt", _lstm_bptt_cuda), ("gru_recurrence", _gru_recurrence_cuda), ("gru_gate", _gru_gate_cuda),
                   ("gru_recurrence_train", _gru_recurrence_train_cuda), ("gru_bptt", _gru_bptt_cuda), ("dwconv2d_wgrad", _dwconv2d_wgrad_cuda),
                   ("blocked_from_nchw", _blocked_from_nchw_cuda), ("nchw_from_blocked", _nchw_from_blocked_cuda),
                   ("conv2d_b16_fwd", _conv2d_b16_fwd_cuda), ("conv2d_b16_head_fwd", _conv2d_b16_head_fwd_cuda),
                   ("conv2d_b16_nchw_fwd", _conv2d_b16_nchw_fwd_cuda), ("blocked_from_nchw_sum", _blocked_from_nchw_sum_cuda)):
    _LIBRARY.impl(_name, _fn, "CUDA")


@torch.library.register_fake("unetmi::bn_train_stats")
def _(z, eps, momentum, running_mean, running_var):
    return z.new_empty((z.shape[1],)), z.new_empty((z.shape[1],))


@torch.library.register_fake("unetmi::bn_batch_stats")
def _(z, eps):
    return z.new_empty((z.shape[1],)), z.new_empty((z.shape[1],))


@torch.library.register_fake("unetmi::bn_running_update")
def _(running_mean, running_var, mean, invstd, count, eps, momentum):
    return None


@torch.library.register_fake("unetmi::bn_act_fwd")
def _(z, mean, invstd, gamma, beta, act, act_param):
    return torch.empty_like(z)


@torch.library.register_fake("unetmi::bn_act_bwd")
def _(dy, z, mean, invstd, gamma, beta, act, act_param, batch_stats):
    c = z.new_empty((z.shape[1],))
    return torch.empty_like(z), c, torch.empty_like(c), torch.empty_like(c)


@torch.library.register_fake("unetmi::conv2d_wgrad")
def _(in0, in1, dout, geom, flags):
    C = in0.shape[1] + (in1.shape[1] if in1 is not None else 0)
    return in0.new_empty(_wgrad_shape(C, geom, flags[0]))


@torch.library.register_fake("unetmi::adam_step")
def _(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    return None


@torch.library.register_fake("unetmi::gather_f32")
def _(src, map):
    return src.new_empty((map.numel(),))


@torch.library.register_fake("unetmi::lstm_recurrence")
def _(wrec, gx, hfull, xcd, channel_major, f16=False):
    return gx.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::lstm_recurrence_train")
def _(wrec, gx, hfull, cell):
    return gx.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::lstm_bptt")
def _(wrec, gates, cell, dhfull):
    return gates.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::dwconv2d_wgrad")
def _(x, x2, dout, kernel, padding, dilation):
    return x.new_empty((x.shape[1], 1, kernel[0], kernel[1])), x.new_empty((x.shape[1],))


@torch.library.register_fake("unetmi::gru_recurrence")
def _(whh, bhh, gx, hfull, channel_major):
    return gx.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::gru_gate")
def _(gx, step, direction, gh, bhh, y, prev_step):
    return None


@torch.library.register_fake("unetmi::gru_recurrence_train")
def _(whh, bhh, gx, hfull, gates):
    return gx.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::gru_bptt")
def _(whh, gates, hfull, dy):
    return gates.new_empty((hfull.shape[1], hfull.shape[2])), gates.new_empty((), dtype=torch.int32)


@torch.library.register_fake("unetmi::blocked_from_nchw")
def _(x):
    N, C, H, W = x.shape
    return x.new_empty((N, blocks_for(C), H, W, 8), dtype=torch.bfloat16)


@torch.library.register_fake("unetmi::nchw_from_blocked")
def _(data, C, f32):
    N, _, H, W, _ = data.shape
    return data.new_empty((N, C, H, W), dtype=torch.float32 if f32 else torch.bfloat16)


@torch.library.register_fake("unetmi::conv2d_b16_fwd")
def _(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, geom, flags):
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    N, _, H, W, _ = x.shape
    OH, OW = conv_out_size(H, kh, sh, ph, dh, oph, flags[0]), conv_out_size(W, kw, sw, pw, dw, opw, flags[0])
    if flags[4]:
        return x.new_empty((N, K, OH, OW), dtype=torch.float32)
    return x.new_empty((N, blocks_for(K), OH, OW, 8), dtype=torch.float16 if len(flags) > 5 and flags[5] else torch.bfloat16)


@torch.library.register_fake("unetmi::blocked_from_nchw_sum")
def _(x, x2, f16=False):
    N, C, H, W = x.shape
    return x.new_empty((N, blocks_for(C), H, W, 8), dtype=torch.float16 if f16 else torch.bfloat16)


@torch.library.register_fake("unetmi::conv2d_b16_nchw_fwd")
def _(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, residual, geom, flags):
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    N, _, H, W, _ = x.shape
    OH, OW = conv_out_size(H, kh, sh, ph, dh, oph, flags[0]), conv_out_size(W, kw, sw, pw, dw, opw, flags[0])
    return x.new_empty((N, K, OH, OW), dtype=torch.float32)


@torch.library.register_fake("unetmi::conv2d_b16_head_fwd")
def _(x, C0, skip, C1, wpacked, bias, scale, shift, act_param, gate, head_w, head_b, geom, flags):
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    N, _, H, W, _ = x.shape
    OH, OW = conv_out_size(H, kh, sh, ph, dh, oph, flags[0]), conv_out_size(W, kw, sw, pw, dw, opw, flags[0])
    P = head_w.shape[0]
    return x.new_empty((N, OH, OW, P) if flags[5] else (N, P, OH, OW), dtype=torch.float32)


bn_train_stats = torch.ops.unetmi.bn_train_stats
bn_batch_stats = torch.ops.unetmi.bn_batch_stats
bn_running_update = torch.ops.unetmi.bn_running_update
bn_act_fwd = torch.ops.unetmi.bn_act_fwd
bn_act_bwd = torch.ops.unetmi.bn_act_bwd
conv2d_wgrad = torch.ops.unetmi.conv2d_wgrad
adam_step = torch.ops.unetmi.adam_step
gather_f32 = torch.ops.unetmi.gather_f32
lstm_recurrence = torch.ops.unetmi.lstm_recurrence
lstm_recurrence_train = torch.ops.unetmi.lstm_recurrence_train
lstm_bptt = torch.ops.unetmi.lstm_bptt
gru_recurrence = torch.ops.unetmi.gru_recurrence
gru_recurrence_train = torch.ops.unetmi.gru_recurrence_train
gru_gate = torch.ops.unetmi.gru_gate
gru_bptt = torch.ops.unetmi.gru_bptt
dwconv2d_wgrad = torch.ops.unetmi.dwconv2d_wgrad
blocked_from_nchw = torch.ops.unetmi.blocked_from_nchw
nchw_from_blocked = torch.ops.unetmi.nchw_from_blocked
conv2d_b16_fwd = torch.ops.unetmi.conv2d_b16_fwd
conv2d_b16_head_fwd = torch.ops.unetmi.conv2d_b16_head_fwd
conv2d_b16_nchw_fwd = torch.ops.unetmi.conv2d_b16_nchw_fwd
blocked_from_nchw_sum = torch.ops.unetmi.blocked_from_nchw_sum
