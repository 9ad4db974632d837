"""Train-mode forward/backward of the fused blocks, the flat-bucket Adam and the
data-parallel training step.

The reference trains its modules with stock autograd (there is no trainer in the
repository); here every Encoder / Decoder block in ``train()`` mode is one
dispatcher op (``torch.ops.unetmi.conv_norm_act``, backward registered with
``torch.library.register_autograd``) whose forward and backward are HIP launches:

    forward   conv (unetmi_conv2d_fwd, no epilogue) -> batch statistics
              (unetmi_bn_train_stats) -> normalise + activation (unetmi_bn_act_fwd)
              [-> running statistics (unetmi_bn_running_update)]
    backward  unetmi_bn_act_bwd -> weight gradient (unetmi_conv2d_wgrad) and input
              gradient (unetmi_conv2d_fwd on the gradient with the transposed flag
              flipped: same weight tensor, conv <-> transposed conv)

Autograd itself only routes tensors between blocks (skip connections, gradient
accumulation).  Parameters and gradients can be flattened into single buckets
(``FlatAdam``), so the data-parallel exchange is ONE all-reduce over RCCL and the
optimizer is one kernel.
"""

import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .ops import conv2d_fwd, conv_out_size, make_desc, pack_conv_weights

from . import ops as _ops


def _dense(t, what):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise ValueError("{} must be a contiguous fp32 device tensor".format(what))


# ---------------------------------------------------------------------------------------
# the training entry points: torch.ops.unetmi.* (registered in ops.py), under the names this module's callers use
# ---------------------------------------------------------------------------------------
def bn_train_stats(z, eps, momentum, running_mean, running_var):
    """(mean, invstd) of z[N,C,...] over all but the channel axis; running stats updated in place."""
    return _ops.bn_train_stats(z, eps, momentum, running_mean, running_var)


def bn_act_fwd(z, mean, invstd, gamma, beta, act, act_param):
    return _ops.bn_act_fwd(z, mean, invstd, gamma, beta, act, act_param)


def bn_act_bwd(dy, z, mean, invstd, gamma, beta, act, act_param, batch_stats):
    """(dz, dgamma, dbeta, dbias).  dy may be a channel slice of a wider contiguous tensor."""
    return _ops.bn_act_bwd(dy, z, mean, invstd, gamma, beta, act, act_param, bool(batch_stats))


def conv2d_wgrad(desc, in0, in1, dout):
    """Weight gradient in torch layout for the layer ``desc`` describes."""
    geom = [desc.K, desc.kh, desc.kw, desc.sh, desc.sw, desc.ph, desc.pw, desc.dh, desc.dw, desc.oph, desc.opw]
    return _ops.conv2d_wgrad(in0, in1, dout, geom, [int(desc.transposed), int(desc.in_mode)])


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    _ops.adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step)
    torch.autograd.graph.increment_version(param)      # written through a raw pointer


# ---------------------------------------------------------------------------------------
# packing weights on the device (parameters change every step)
# ---------------------------------------------------------------------------------------
_pack_index_cache = {}


def device_pack_index(cin, geom, transposed, device, c1=0):
    """int32 gather map of unetmi_conv2d_pack_weights for unetmi_gather_f32: packed[i] = weight.flatten()[map[i]],
    0 where map[i] is -1.
    The pack is a permutation with zero padding, so the map is found once per layer shape by
    packing the element numbers on the host."""
    key = (cin, c1, tuple(geom), int(transposed), str(device))
    idx = _pack_index_cache.get(key)
    if idx is None:
        K, kh, kw = geom[0], geom[1], geom[2]
        shape = (cin, K, kh, kw) if transposed else (K, cin, kh, kw)
        numel = cin * K * kh * kw
        if numel >= (1 << 24):
            raise ValueError("weight tensor too large for the float-coded index map")
        numbered = torch.arange(1, numel + 1, dtype=torch.float32).reshape(shape)
        packed = pack_conv_weights(numbered, cin - c1, c1, geom, [int(transposed), _lib.IN_SINGLE, _lib.ACT_NONE, 0])
        idx = (packed.round().long() - 1).to(torch.int32).to(device)         # -1 = padding
        _pack_index_cache[key] = idx
    return idx


_const_cache = {}


def _constant(value, n, device):
    """A cached fp32 vector of n copies of value (never written: epilogue identity parameters)."""
    key = (float(value), int(n), str(device))
    t = _const_cache.get(key)
    if t is None:
        t = _const_cache[key] = torch.full((n,), float(value), dtype=torch.float32, device=device)
    return t


def pack_on_device(weight, geom, transposed, c1=0):
    """c1: channels of the second tensor when the layer reads a concatenation (the kernel chunks
    the channels per source, so the image depends on the split)."""
    cin = weight.shape[0] if transposed else weight.shape[1]
    idx = device_pack_index(cin, geom, transposed, weight.device, c1)
    return _ops.gather_f32(weight.detach().contiguous(), idx)        # one launch; padding positions are -1 in the map


# Batch statistics from the convolution's epilogue (unetmi_conv2d_fwd_stats) where the layer's kernel has one, instead
# of the separate statistics pass (unetmi_bn_train_stats).  Off by default: at the benchmark's batch of 8 the pass reads
# a tensor the convolution has just left in the last-level cache, and the step time is the same either way
# (19.62 ms with the epilogue, 19.58 without), while the epilogue's fp32 tile sums are slightly less exact than the
# pass's double sums.  UNETMI_STATS_EPILOGUE=1 (or setting this flag) turns it on.
STATS_IN_EPILOGUE = os.environ.get("UNETMI_STATS_EPILOGUE", "0") == "1"


def _plain_conv(x, x2, weight, bias, geom, transposed, stats_eps=None):
    """conv / transposed conv + bias, nothing else.  With ``stats_eps``: (z, mean, invstd), the batch statistics of z
    taken in the convolution's epilogue where the layer's kernel has that epilogue, by a pass over z otherwise."""
    K = geom[0]
    wpacked = pack_on_device(weight, geom, transposed, x2.shape[1] if x2 is not None else 0)
    ones, zeros = _constant(1.0, K, x.device), _constant(0.0, K, x.device)
    b = bias.detach() if bias is not None else zeros
    mode = _lib.IN_SINGLE if x2 is None else _lib.IN_CONCAT
    flags = [int(transposed), mode, _lib.ACT_NONE, 0]
    if stats_eps is not None and STATS_IN_EPILOGUE and _ops.conv2d_fwd_stats_supported(x, x2, geom, flags):
        try:
            return _ops.conv2d_fwd_stats(x, x2, wpacked, b, ones, zeros, stats_eps, geom, flags)
        except _lib.UnetmiError as e:
            if e.code != _lib.E_UNSUPPORTED:
                raise
    try:
        z = conv2d_fwd(x, x2, wpacked, b, ones, zeros, None, None, None, geom, flags)
    except _lib.UnetmiError as e:
        if e.code != _lib.E_UNSUPPORTED or x2 is None:
            raise
        z = conv2d_fwd(torch.cat([x, x2], dim=1), None, pack_on_device(weight, geom, transposed), b, ones, zeros,
                       None, None, None, geom, [int(transposed), _lib.IN_SINGLE, _lib.ACT_NONE, 0])
    if stats_eps is None:
        return z
    mean, invstd = _ops.bn_batch_stats(z, stats_eps)
    return z, mean, invstd


def _input_gradient(dz, weight, geom, transposed, in_hw):
    """Gradient w.r.t. the (concatenated) input: the opposite kind of convolution of dz with the
    same weight tensor.  geom/transposed describe the FORWARD layer, in_hw its input size."""
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    cin = weight.shape[0] if transposed else weight.shape[1]
    H, W = in_hw
    if transposed:
        g2 = [cin, kh, kw, sh, sw, ph, pw, dh, dw, 0, 0]          # plain conv, weight read as [K'=Cin][C'=Cout]
    else:
        OH, OW = dz.shape[2], dz.shape[3]
        g2 = [cin, kh, kw, sh, sw, ph, pw, dh, dw,
              H - conv_out_size(OH, kh, sh, ph, dh, 0, True), W - conv_out_size(OW, kw, sw, pw, dw, 0, True)]
    dx = _plain_conv(dz, None, weight, None, g2, not transposed)
    if tuple(dx.shape[2:]) != (H, W):
        raise RuntimeError("input-gradient geometry mismatch: {} vs {}".format(tuple(dx.shape[2:]), (H, W)))
    return dx


# ---------------------------------------------------------------------------------------
# unetmi::conv_norm_act -- the differentiable fused block as ONE dispatcher op:
#   y = act(norm(conv(cat(x, skip))))     norm_mode 0: none, 1: BatchNorm2d with batch statistics,
#                                          2: BatchNorm2d with its running statistics (still differentiable)
# Functional (it returns the batch statistics instead of updating anything), so its backward is registered with
# torch.library.register_autograd; the running-statistics update is a separate op (bn_running_update).
# Outputs: y, and what the backward needs -- z (the convolution output), mean, invstd.
# ---------------------------------------------------------------------------------------
_ops._LIBRARY.define(
    "conv_norm_act(Tensor x, Tensor? skip, Tensor weight, Tensor? bias, Tensor? gamma, Tensor? beta, "
    "Tensor? running_mean, Tensor? running_var, int[] geom, bool transposed, int norm_mode, float eps, int act, "
    "Tensor? act_param) -> (Tensor, Tensor, Tensor, Tensor)")


def _conv_norm_act_cuda(x, skip, weight, bias, gamma, beta, running_mean, running_var, geom, transposed, norm_mode,
                        eps, act, act_param):
    if norm_mode == 1:
        z, mean, invstd = _plain_conv(x, skip, weight, bias, geom, transposed, stats_eps=eps)
    else:
        z = _plain_conv(x, skip, weight, bias, geom, transposed)
    K, dev = geom[0], z.device
    if norm_mode == 1:
        pass
    elif norm_mode == 2:
        mean = running_mean.detach()
        invstd = torch.rsqrt(running_var.detach() + eps)
    else:
        mean, invstd = _constant(0.0, K, dev), _constant(1.0, K, dev)
    g = gamma.detach() if gamma is not None else _constant(1.0, K, dev)
    b = beta.detach() if beta is not None else _constant(0.0, K, dev)
    y = _ops.bn_act_fwd(z, mean, invstd, g, b, act, act_param)
    # mean / invstd may be cached constants or views of module buffers: outputs of an op must be its own tensors
    if norm_mode != 1:
        mean, invstd = mean.clone(), invstd.clone()
    return y, z, mean, invstd


_ops._LIBRARY.impl("conv_norm_act", _conv_norm_act_cuda, "CUDA")


@torch.library.register_fake("unetmi::conv_norm_act")
def _(x, skip, weight, bias, gamma, beta, running_mean, running_var, geom, transposed, norm_mode, eps, act, act_param):
    N, _, H, W = x.shape
    K, kh, kw, sh, sw, ph, pw, dh, dw, oph, opw = geom
    z = x.new_empty((N, K, conv_out_size(H, kh, sh, ph, dh, oph, transposed), conv_out_size(W, kw, sw, pw, dw, opw, transposed)))
    return torch.empty_like(z), z, x.new_empty((K,)), x.new_empty((K,))


def _conv_norm_act_setup(ctx, inputs, output):
    x, skip, weight, bias, gamma, beta, _, _, geom, transposed, norm_mode, _, act, act_param = inputs
    _, z, mean, invstd = output
    ctx.geom, ctx.transposed, ctx.norm_mode, ctx.act = list(geom), bool(transposed), norm_mode, act
    ctx.in_hw, ctx.c0 = tuple(x.shape[2:]), x.shape[1]
    ctx.has = (skip is not None, bias is not None, gamma is not None, beta is not None, act_param is not None)
    ctx.set_materialize_grads(False)           # z / mean / invstd are outputs only so that they can be saved
    ctx.save_for_backward(x, skip, weight, z, mean, invstd, gamma, beta, act_param)


def _conv_norm_act_backward(ctx, dy, dz_unused, dmean_unused, dinvstd_unused):
    x, skip, weight, z, mean, invstd, gamma, beta, act_param = ctx.saved_tensors
    none = (None,) * 8
    if dy is None:
        return (None,) * 6 + none
    geom, transposed = ctx.geom, ctx.transposed
    need = ctx.needs_input_grad
    K, dev = geom[0], z.device
    g = gamma.detach() if gamma is not None else _constant(1.0, K, dev)
    b = beta.detach() if beta is not None else _constant(0.0, K, dev)
    dz, dgamma, dbeta, dbias = bn_act_bwd(dy, z, mean, invstd, g, b, ctx.act, act_param, ctx.norm_mode == 1)
    dweight = None
    if need[2]:
        N, C0, H, W = x.shape
        C1 = skip.shape[1] if skip is not None else 0
        d = make_desc(N, C0, C1, H, W, geom, [int(transposed), _lib.IN_CONCAT if C1 else _lib.IN_SINGLE, 0, 0])
        if transposed or skip is None:
            dweight = conv2d_wgrad(d, x, skip, dz)
        else:                                   # plain conv over a concatenation: not a case these models have
            d = make_desc(N, C0 + C1, 0, H, W, geom, [0, _lib.IN_SINGLE, 0, 0])
            dweight = conv2d_wgrad(d, torch.cat([x, skip], 1), None, dz)
    dx = dskip = None
    if need[0] or (skip is not None and need[1]):
        dcat = _input_gradient(dz, weight, geom, transposed, ctx.in_hw)
        if skip is None:
            dx = dcat
        else:
            dx, dskip = dcat[:, :ctx.c0], dcat[:, ctx.c0:]
    return (dx if need[0] else None, dskip if (ctx.has[0] and need[1]) else None, dweight,
            dbias if (ctx.has[1] and need[3]) else None, dgamma if (ctx.has[2] and need[4]) else None,
            dbeta if (ctx.has[3] and need[5]) else None) + none


torch.library.register_autograd("unetmi::conv_norm_act", _conv_norm_act_backward, setup_context=_conv_norm_act_setup)


def conv_norm_act(conv, norm, act_code, act_param, x, skip=None):
    """Differentiable fused block for nn.Conv2d / nn.ConvTranspose2d ``conv`` followed by ``norm``
    (nn.BatchNorm2d, nn.Identity or None) and an activation given by its epilogue code: one
    ``torch.ops.unetmi.conv_norm_act`` node, then the BatchNorm bookkeeping that a train-mode forward does."""
    from .fused import FusedConv
    geom, transposed = FusedConv(conv).geometry()
    bn = norm if isinstance(norm, nn.BatchNorm2d) else None
    if norm is not None and bn is None and not isinstance(norm, nn.Identity):
        raise NotImplementedError("train-mode forward is built for BatchNorm2d blocks only (got {})".format(
            type(norm).__name__))
    if conv.groups != 1 or (isinstance(conv, nn.Conv2d) and conv.padding_mode != "zeros"):
        raise NotImplementedError("train-mode forward: grouped or non-zero-padded convolutions are not built")
    batch_stats = bn is not None and (bn.training or bn.running_mean is None)
    if bn is not None and batch_stats and (bn.momentum is None or not bn.track_running_stats):
        raise NotImplementedError("train-mode BatchNorm2d needs a numeric momentum and tracked running statistics")
    mode = 0 if bn is None else (1 if batch_stats else 2)
    y, z, mean, invstd = torch.ops.unetmi.conv_norm_act(
        x, skip, conv.weight, conv.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
        bn.running_mean if mode == 2 else None, bn.running_var if mode == 2 else None, list(geom), bool(transposed),
        mode, bn.eps if bn is not None else 0.0, act_code, act_param)
    if mode == 1:
        _bn_bookkeeping(bn, z, mean, invstd)
    return y


# ---------------------------------------------------------------------------------------
# unetmi::bn_act -- BatchNorm2d + activation of a tensor that is already there (the residual block normalises AFTER
# its PReLU, UNet_m.py:545-551; a block with dropout normalises the dropped-out convolution): the norm half of
# conv_norm_act as a functional op of its own, same kernels, same modes.
# ---------------------------------------------------------------------------------------
_ops._LIBRARY.define(
    "bn_act(Tensor z, Tensor? gamma, Tensor? beta, Tensor? running_mean, Tensor? running_var, int norm_mode, float eps, "
    "int act, Tensor? act_param) -> (Tensor, Tensor, Tensor)")


def _bn_act_cuda(z, gamma, beta, running_mean, running_var, norm_mode, eps, act, act_param):
    z = z.contiguous()
    K, dev = z.shape[1], z.device
    if norm_mode == 1:
        mean, invstd = _ops.bn_batch_stats(z, eps)
    else:
        mean = running_mean.detach().clone()
        invstd = torch.rsqrt(running_var.detach() + eps)
    g = gamma.detach() if gamma is not None else _constant(1.0, K, dev)
    b = beta.detach() if beta is not None else _constant(0.0, K, dev)
    return _ops.bn_act_fwd(z, mean, invstd, g, b, act, act_param), mean, invstd


_ops._LIBRARY.impl("bn_act", _bn_act_cuda, "CUDA")


@torch.library.register_fake("unetmi::bn_act")
def _(z, gamma, beta, running_mean, running_var, norm_mode, eps, act, act_param):
    return torch.empty_like(z), z.new_empty((z.shape[1],)), z.new_empty((z.shape[1],))


def _bn_act_setup(ctx, inputs, output):
    z, gamma, beta, _, _, norm_mode, _, act, act_param = inputs
    _, mean, invstd = output
    ctx.norm_mode, ctx.act = norm_mode, act
    ctx.has = (gamma is not None, beta is not None)
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(z, mean, invstd, gamma, beta, act_param)


def _bn_act_backward(ctx, dy, dmean_unused, dinvstd_unused):
    z, mean, invstd, gamma, beta, act_param = ctx.saved_tensors
    if dy is None:
        return (None,) * 9
    z = z.contiguous()
    K, dev = z.shape[1], z.device
    g = gamma.detach() if gamma is not None else _constant(1.0, K, dev)
    b = beta.detach() if beta is not None else _constant(0.0, K, dev)
    dz, dgamma, dbeta, _ = bn_act_bwd(dy.contiguous(), z, mean, invstd, g, b, ctx.act, act_param, ctx.norm_mode == 1)
    need = ctx.needs_input_grad
    return (dz if need[0] else None, dgamma if (ctx.has[0] and need[1]) else None,
            dbeta if (ctx.has[1] and need[2]) else None) + (None,) * 6


torch.library.register_autograd("unetmi::bn_act", _bn_act_backward, setup_context=_bn_act_setup)


def _bn_bookkeeping(bn, z, mean, invstd):
    """What a train-mode BatchNorm2d forward does beside normalising: running statistics and the batch counter."""
    with torch.no_grad():
        count = z.numel() // z.shape[1]
        _ops.bn_running_update(bn.running_mean, bn.running_var, mean, invstd, count, bn.eps, bn.momentum)
        torch.autograd.graph.increment_version(bn.running_mean)      # written through raw pointers
        torch.autograd.graph.increment_version(bn.running_var)
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1


def batch_norm_act(bn, z, act_code=_lib.ACT_NONE, act_param=None):
    """Differentiable nn.BatchNorm2d ``bn`` (+ an activation by its epilogue code) of ``z`` on the library's kernels:
    batch statistics and bookkeeping in train mode, the running statistics in eval mode."""
    batch_stats = bn.training or bn.running_mean is None
    if batch_stats and (bn.momentum is None or not bn.track_running_stats):
        raise NotImplementedError("train-mode BatchNorm2d needs a numeric momentum and tracked running statistics")
    mode = 1 if batch_stats else 2
    y, mean, invstd = torch.ops.unetmi.bn_act(z, bn.weight, bn.bias, bn.running_mean if mode == 2 else None,
                                              bn.running_var if mode == 2 else None, mode, bn.eps, act_code, act_param)
    if mode == 1:
        _bn_bookkeeping(bn, z, mean, invstd)
    return y


# ---------------------------------------------------------------------------------------
# autograd of unetmi::dwconv2d_fwd (the depthwise 'same' convolution of a residual-block layer): the input gradient
# is the same kernel on the output gradient with each channel's taps reversed, the weight / bias gradient is
# unetmi_dwconv2d_wgrad.
# ---------------------------------------------------------------------------------------
def _dwconv_setup(ctx, inputs, output):
    x, x2, weight, bias, padding, dilation = inputs
    ctx.padding, ctx.dilation = list(padding), list(dilation)
    ctx.has = (x2 is not None, bias is not None)
    ctx.save_for_backward(x, x2, weight)


def _dwconv_backward(ctx, dy):
    x, x2, weight = ctx.saved_tensors
    need = ctx.needs_input_grad
    dy = dy.contiguous()
    dx = dweight = dbias = None
    if need[0] or (ctx.has[0] and need[1]):
        dx = _ops.dwconv2d_fwd(dy, None, weight.detach().flip(2, 3).contiguous(), None, ctx.padding, ctx.dilation)
    if need[2] or (ctx.has[1] and need[3]):
        dweight, dbias = _ops.dwconv2d_wgrad(x, x2, dy, [weight.shape[2], weight.shape[3]], ctx.padding, ctx.dilation)
    return (dx if need[0] else None, dx if (ctx.has[0] and need[1]) else None, dweight if need[2] else None,
            dbias if (ctx.has[1] and need[3]) else None, None, None)


torch.library.register_autograd("unetmi::dwconv2d_fwd", _dwconv_backward, setup_context=_dwconv_setup)


# ---------------------------------------------------------------------------------------
# optimizer and data-parallel step
# ---------------------------------------------------------------------------------------
class FlatBucket:
    """Parameters and their .grad as views of two contiguous fp32 buffers, so the data-parallel
    gradient exchange is a single all-reduce.  Device-agnostic (the host logic is exercised with
    gloo on CPU tensors); the optimizer on top of it is device-only."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatBucket: no parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatBucket: parameters must be fp32 tensors on one device")
        total = sum(p.numel() for p in self.params)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.flat)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                off += n

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:                       # re-attach if something replaced .grad
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)
            off += n

    def detach_grads(self):
        """Before a backward pass whose gradients collect_grads() will gather: with .grad unset, autograd hands
        each parameter its gradient tensor as it is, instead of launching one small add per parameter into the
        bucket's views (81 launches per UNet step)."""
        for p in self.params:
            p.grad = None

    def collect_grads(self):
        """Copy the gradients autograd left on the parameters into the bucket (a multi-tensor copy: a few
        launches for all of them), zero the slices of parameters that received none, and point every .grad at
        its slice of the bucket again."""
        srcs, dsts, off = [], [], 0
        for p in self.params:
            n = p.numel()
            view = self.grad[off:off + n].view(p.shape)
            g = p.grad
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                srcs.append(g if g.dtype == torch.float32 else g.float())
                dsts.append(view)
            p.grad = view
            off += n
        if srcs:
            with torch.no_grad():
                torch._foreach_copy_(dsts, srcs)

    def all_reduce(self, group=None):
        """Sum the gradient bucket over the data-parallel group (RCCL on GPUs); returns the group
        size, by which the caller divides (FlatAdam.step folds the division into the update)."""
        import torch.distributed as dist
        from . import comm
        if self.grad.is_cuda and group is None and comm.world() > 0:
            # the library's own communicator (unetmi_comm_init): one ncclAllReduce on the current stream
            comm.all_reduce_sum_(self.grad)
            return comm.world()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "gloo" and self.grad.is_cuda and os.environ.get("UNETMI_GLOO_DEVICE_TENSORS") != "1":
                # gloo stages device tensors through the host on its own streams; do the staging here,
                # synchronously, so the exchange cannot interleave with this process's queued kernels
                host = self.grad.cpu()
                dist.all_reduce(host, group=group)
                self.grad.copy_(host)
            else:
                dist.all_reduce(self.grad, group=group)
            return dist.get_world_size(group)
        return 1


class FlatAdam(FlatBucket):
    """torch.optim.Adam semantics (lr, betas, eps, weight_decay; no amsgrad) over the flat bucket:
    the update of all parameters is one kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params)
        if self.flat.device.type != "cuda":
            raise ValueError("FlatAdam: parameters must live on the GPU")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0

    def step(self, world_size=1):
        self.step_count += 1
        adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                  self.eps, self.weight_decay, 1.0 / world_size, self.step_count)
        for p in self.params:                      # parameters keep their own version counters
            torch.autograd.graph.increment_version(p)


def train_step(model, optimizer, x, target, group=None):
    """One data-parallel step: train-mode forward, loss mean((mask - target)^2), backward,
    all-reduce(mean) of the gradient bucket, Adam.  Returns the local loss (a device scalar)."""
    optimizer.detach_grads()
    loss = torch.mean((model(x) - target) ** 2)
    loss.backward()
    optimizer.collect_grads()
    world = optimizer.all_reduce(group)
    optimizer.step(world)
    return loss.detach()
