"""Training-step kernels against torch CPU autograd (the arithmetic the reference trains with) and
against the golden training step recorded from the reference."""

import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.util import assert_close, load_golden, sample_index

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


@pytest.mark.parametrize("act", ["leaky", "relu", "silu", "none"])
def test_bn_train_forward_backward(device, act):
    from unet_amd import _lib
    from unet_amd.train import bn_act_bwd, bn_act_fwd, bn_train_stats
    code = dict(leaky=_lib.ACT_LEAKY, relu=_lib.ACT_RELU, silu=_lib.ACT_SILU, none=_lib.ACT_NONE)[act]
    fn = dict(leaky=lambda t: F.leaky_relu(t, 0.01), relu=F.relu, silu=F.silu, none=lambda t: t)[act]
    N, C, H, W = 3, 5, 17, 29
    z = (_rand((N, C, H, W), 1) * 2 + 0.7).requires_grad_()
    bn = nn.BatchNorm2d(C).train()
    with torch.no_grad():
        bn.weight.copy_(_rand((C,), 2).abs() + 0.5)
        bn.bias.copy_(_rand((C,), 3))
    dy_wide = _rand((N, C + 3, H, W), 4)
    dy = dy_wide[:, 2:2 + C]
    y = fn(bn(z))
    y.backward(dy)
    rm, rv = torch.zeros(C, device=device), torch.ones(C, device=device)
    zd = z.detach().to(device)
    mean, invstd = bn_train_stats(zd, bn.eps, bn.momentum, rm, rv)
    par = torch.full((1,), 0.01, device=device) if act == "leaky" else None
    gam, bet = bn.weight.detach().to(device), bn.bias.detach().to(device)
    yd = bn_act_fwd(zd, mean, invstd, gam, bet, code, par)
    assert_close(yd, y, tol=1e-5, what="bn+act forward")
    assert_close(rm, bn.running_mean, tol=1e-5, what="running_mean")
    assert_close(rv, bn.running_var, tol=1e-5, what="running_var")
    dz, dgamma, dbeta, dbias = bn_act_bwd(dy_wide.to(device)[:, 2:2 + C], zd, mean, invstd, gam, bet, code, par, True)
    assert_close(dz, z.grad, tol=1e-4, what="dz")
    assert_close(dgamma, bn.weight.grad, tol=1e-5, what="dgamma")
    assert_close(dbeta, bn.bias.grad, tol=1e-5, what="dbeta")
    assert dbias.abs().max().item() <= 1e-4 * dy.abs().sum().item() / C      # sums to zero up to rounding


WGRAD_CASES = [
    # transposed, C0, C1, K, H, W, k, s, p, op
    (0, 6, 0, 10, 13, 37, (7, 5), (2, 2), (3, 2), (0, 0)),
    (0, 64, 0, 64, 9, 40, (7, 5), (2, 1), (3, 2), (0, 0)),
    (0, 32, 0, 64, 5, 70, (1, 7), (1, 1), (0, 3), (0, 0)),
    (0, 2, 0, 32, 20, 33, (7, 1), (1, 1), (3, 0), (0, 0)),
    (0, 64, 0, 64, 11, 33, (5, 3), (2, 2), (2, 1), (0, 0)),
    (0, 1, 0, 1, 9, 45, (1, 1), (1, 1), (0, 0), (0, 0)),
    (1, 64, 64, 64, 5, 19, (7, 5), (2, 2), (3, 2), (0, 0)),
    (1, 64, 64, 64, 4, 32, (5, 3), (2, 2), (2, 1), (0, 1)),
    (1, 20, 13, 7, 6, 35, (5, 3), (2, 1), (2, 1), (0, 0)),
    (1, 64, 32, 1, 12, 33, (7, 1), (1, 1), (3, 0), (0, 0)),
    (1, 64, 64, 64, 6, 70, (1, 7), (1, 1), (0, 3), (0, 0)),
    (1, 64, 0, 64, 2, 32, (5, 3), (2, 1), (2, 1), (0, 0)),
    # thin layers on the streaming kernel: several row chunks and column strips, 3-row kernel, one and two thin channels
    (1, 8, 3, 2, 70, 130, (3, 1), (1, 1), (1, 0), (0, 0)),
    (0, 1, 0, 20, 100, 65, (7, 1), (1, 1), (3, 0), (0, 0)),
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "{}{}+{}-{}k{}x{}s{}x{}".format(
    "T" if c[0] else "C", c[1], c[2], c[3], c[6][0], c[6][1], c[7][0], c[7][1]))
def test_wgrad_and_input_gradient(device, case):
    from unet_amd import _lib
    from unet_amd.ops import make_desc
    from unet_amd.train import _input_gradient, conv2d_wgrad
    transposed, C0, C1, K, H, W, k, s, p, op = case
    N = 3
    cin = C0 + C1
    x = _rand((N, cin, H, W), 11).requires_grad_()
    if transposed:
        conv = nn.ConvTranspose2d(cin, K, k, s, p, output_padding=op)
    else:
        conv = nn.Conv2d(cin, K, k, s, p)
    y = conv(x)
    dy = _rand(tuple(y.shape), 12)
    y.backward(dy)
    geom = [K, k[0], k[1], s[0], s[1], p[0], p[1], 1, 1, op[0], op[1]]
    d = make_desc(N, C0, C1, H, W, geom, [transposed, _lib.IN_CONCAT if C1 else _lib.IN_SINGLE, 0, 0])
    xd = x.detach().to(device)
    in0 = xd[:, :C0].contiguous()
    in1 = xd[:, C0:].contiguous() if C1 else None
    dyd = dy.to(device)
    dw = conv2d_wgrad(d, in0, in1, dyd)
    assert_close(dw, conv.weight.grad, tol=2e-5, what="weight gradient")
    dw2 = conv2d_wgrad(d, in0, in1, dyd)
    assert torch.equal(dw, dw2), "weight gradient is not reproducible"
    dx = _input_gradient(dyd, conv.weight.detach().to(device), geom, bool(transposed), (H, W))
    assert_close(dx, x.grad, tol=2e-5, what="input gradient")


@pytest.mark.parametrize("case", WGRAD_CASES + [
    (0, 64, 0, 64, 257, 61, (7, 5), (2, 2), (3, 2), (0, 0)),         # several row tiles, ragged last tile
    (1, 64, 64, 64, 33, 40, (7, 5), (2, 2), (3, 2), (1, 1)),         # two row phases x two column phases
    (0, 128, 0, 128, 9, 70, (5, 3), (2, 1), (2, 1), (0, 0)),         # two output-channel blocks
], ids=lambda c: "{}{}+{}-{}k{}x{}s{}x{}h{}".format("T" if c[0] else "C", c[1], c[2], c[3], c[6][0], c[6][1], c[7][0], c[7][1], c[4]))
def test_batch_statistics_from_the_conv_epilogue(device, case):
    """unetmi_conv2d_fwd_stats (train() mode of nn.Conv2d -> nn.BatchNorm2d, UNet_m.py:253-256): the convolution output
    is bit-identical to unetmi_conv2d_fwd's, mean and 1/sqrt(var + eps) match float64 statistics of that output, and a
    second call gives the same bits; layers of the streaming kernel report that they have no such epilogue."""
    from unet_amd import _lib, ops
    from unet_amd.train import _constant, pack_on_device
    transposed, C0, C1, K, H, W, k, s_, p_, op = case
    N = 3
    geom = [K, k[0], k[1], s_[0], s_[1], p_[0], p_[1], 1, 1, op[0], op[1]]
    flags = [int(transposed), _lib.IN_CONCAT if C1 else _lib.IN_SINGLE, _lib.ACT_NONE, 0]
    x = _rand((N, C0 + C1, H, W), 41).to(device)
    in0 = x[:, :C0].contiguous()
    in1 = x[:, C0:].contiguous() if C1 else None
    wshape = (C0 + C1, K) + tuple(k) if transposed else (K, C0 + C1) + tuple(k)
    weight = (_rand(wshape, 42) * 0.1).to(device)
    bias = _rand((K,), 43).to(device)
    wp = pack_on_device(weight, geom, bool(transposed), C1)
    ones, zeros = _constant(1.0, K, device), _constant(0.0, K, device)
    z_ref = ops.conv2d_fwd(in0, in1, wp, bias, ones, zeros, None, None, None, geom, flags)
    if not ops.conv2d_fwd_stats_supported(in0, in1, geom, flags):
        name = ops.conv_kernel_name(ops.make_desc(N, C0, C1, H, W, geom, flags))
        assert "direct" in name, name
        with pytest.raises(_lib.UnetmiError):
            ops.conv2d_fwd_stats(in0, in1, wp, bias, ones, zeros, 1e-5, geom, flags)
        return
    z, mean, invstd = ops.conv2d_fwd_stats(in0, in1, wp, bias, ones, zeros, 1e-5, geom, flags)
    assert torch.equal(z, z_ref)
    z64 = z_ref.double()
    m64 = z64.mean(dim=(0, 2, 3))
    v64 = z64.var(dim=(0, 2, 3), unbiased=False)
    assert_close(mean.cpu(), m64.float().cpu(), tol=2e-6, what="batch mean")
    assert_close(invstd.cpu(), torch.rsqrt(v64 + 1e-5).float().cpu(), tol=2e-6, what="1/sqrt(var + eps)")
    z2, mean2, invstd2 = ops.conv2d_fwd_stats(in0, in1, wp, bias, ones, zeros, 1e-5, geom, flags)
    assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2) and torch.equal(z, z2)
    m_pass, i_pass = ops.bn_batch_stats(z_ref, 1e-5)
    assert_close(mean.cpu(), m_pass.cpu(), tol=2e-6, what="batch mean vs the statistics pass")
    assert_close(invstd.cpu(), i_pass.cpu(), tol=2e-6, what="invstd vs the statistics pass")


def test_train_block_with_statistics_epilogue(device, monkeypatch):
    """conv -> BatchNorm2d(batch statistics) -> LeakyReLU as one training node, with the statistics taken in the
    convolution's epilogue (train.STATS_IN_EPILOGUE) and by the separate pass: same output and gradients to 1e-5."""
    from unet_amd import _lib, train
    conv = nn.Conv2d(64, 64, (5, 3), (2, 2), (2, 1)).to(device)
    bn = nn.BatchNorm2d(64).to(device).train()
    x = _rand((4, 64, 65, 48), 51).to(device)
    g = _rand((4, 64, 33, 24), 52).to(device)
    slope = torch.tensor([0.2], device=device)
    results = []
    for flag in (False, True):
        monkeypatch.setattr(train, "STATS_IN_EPILOGUE", flag)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        for p_ in list(conv.parameters()) + list(bn.parameters()):
            p_.grad = None
        xi = x.clone().requires_grad_()
        y = train.conv_norm_act(conv, bn, _lib.ACT_LEAKY, slope, xi)
        y.backward(g)
        results.append([y.detach(), xi.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(),
                        bn.running_var.clone()])
    for a, b, what in zip(results[0], results[1], ("output", "input gradient", "weight gradient", "gamma gradient",
                                                   "beta gradient", "running mean", "running variance")):
        assert_close(b.cpu(), a.cpu(), tol=1e-5, what="statistics epilogue vs statistics pass: " + what)


def test_adam_matches_torch(device):
    from unet_amd.train import adam_step
    n = 100003
    p0, g = _rand((n,), 21), _rand((n,), 22) * 1e-2
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(device)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in (1, 2, 3):
        ref.grad = g * step
        opt.step()
        adam_step(p, (g * step).to(device), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, step)
        # p is bit-rounded at its own magnitude (~1), the update is ~1e-3: compare both at their scale
        assert_close(p, ref.detach(), tol=2e-7, what="parameters after step {}".format(step))
        assert_close(p - p0.to(device), ref.detach() - p0, tol=1e-3, what="update after step {}".format(step))


def _narrow_pair(seed):
    import oracle
    import unet_amd
    ref = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=8))
    unet_amd.seeded_fill_(ref, seed=seed)
    dev = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=8))
    dev.load_state_dict(ref.state_dict(), strict=True)
    return ref, dev


def test_train_step_matches_golden_and_oracle(device):
    """Loss, every gradient, BatchNorm running statistics and parameters after one Adam step of the
    narrow UNet: device vs the oracle run here, and vs the values recorded from the reference."""
    from unet_amd.seeding import normal, uniform
    from unet_amd.train import FlatAdam
    ref, dev = _narrow_pair(11)
    xb = normal("unet_narrow/xb", (2, 2, 257, 29), seed=12)
    target = uniform("unet_narrow/target", (2, 1, 257, 29), 0.0, 1.0, seed=12)
    ref_init = copy.deepcopy(ref)
    ref.train()
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt_ref.zero_grad()
    loss_ref = torch.mean((ref(xb) - target) ** 2)
    loss_ref.backward()
    grads_ref = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    before = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    opt_ref.step()

    dev.to(device).train()
    opt = FlatAdam(dev.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = torch.mean((dev(xb.to(device)) - target.to(device)) ** 2)
    loss.backward()
    golden = load_golden("unet_narrow_trainstep")
    assert abs(loss.item() - float(golden["loss"])) <= 1e-5 * abs(float(golden["loss"]))
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    # Gradients.  The bar is rel-err 1e-4 of the tensor's largest entry where fp32 backprop is that
    # well conditioned; through the 2x32-pixel bottleneck with a batch of 2 the reference's own fp32
    # gradients differ from exact (float64) arithmetic by up to ~6e-3, so there the device is held to
    # three times the reference's own distance from the float64 result.
    ref64 = copy.deepcopy(ref_init).double().train()
    torch.mean((ref64(xb.double()) - target.double()) ** 2).backward()
    exact = {k: p.grad for k, p in ref64.named_parameters()}
    gmax = max(g.abs().max().item() for g in grads_ref.values())
    for k, p in dev.named_parameters():
        g, want = p.grad.detach().cpu(), grads_ref[k]
        scale = max(want.abs().max().item(), 1e-3 * gmax)
        cpu_err = (want.double() - exact[k]).abs().max().item()
        tol = max(1e-4 * scale, 3.0 * cpu_err)
        err = (g.double() - exact[k]).abs().max().item()
        assert err <= tol, "grad {}: err {:.3e} vs tol {:.3e} (scale {:.3e})".format(k, err, tol, scale)
        gs = golden["grad/" + k]
        have = g.reshape(-1)[torch.from_numpy(sample_index(g.numel(), 8))].numpy()
        assert np.abs(have - gs).max() <= tol + cpu_err, "golden grad " + k
    # Optimizer: torch's Adam applied on the CPU to the DEVICE gradients must land on the device's
    # parameters (the first Adam step is lr * g / (|g| + eps), i.e. +-lr with the sign of g, so
    # comparing against the step taken from the reference's own gradients would turn rounding-level
    # differences in near-zero gradients into +-2 lr; gradient parity is established above).
    twin = copy.deepcopy(ref_init).train()
    for (k, p), (_, q) in zip(twin.named_parameters(), dev.named_parameters()):
        p.grad = q.grad.detach().cpu().clone()
    torch.optim.Adam(twin.parameters(), lr=1e-3).step()
    opt.step()
    after_twin, after_ref = twin.state_dict(), ref.state_dict()
    for k, v in dev.state_dict().items():
        have = v.detach().cpu().float()
        if k.endswith("num_batches_tracked"):
            assert int(v.item()) == int(after_ref[k].item()) == 1
            continue
        gs = golden["post/" + k]
        sampled = have.reshape(-1)[torch.from_numpy(sample_index(have.numel(), 8))].numpy()
        if k in grads_ref:
            assert (have - after_twin[k]).abs().max().item() <= 2e-7 * max(have.abs().max().item(), 1.0), "Adam " + k
            assert (have - after_ref[k]).abs().max().item() <= 2.001e-3, "post-step " + k
            assert np.abs(sampled - gs).max() <= 2.001e-3, "golden post-step " + k
        else:                                   # BatchNorm running statistics after one train-mode forward
            assert_close(have, after_ref[k].float(), tol=1e-5, what="post-step " + k)
            assert np.abs(sampled - gs).max() <= 1e-5 * max(np.abs(gs).max(), 1e-6), "golden post-step " + k


def test_eval_after_training_uses_updated_weights(device):
    """The inference kernels cache packed weights; an optimizer step must invalidate them."""
    from unet_amd.seeding import normal, uniform
    from unet_amd.train import FlatAdam, train_step
    ref, dev = _narrow_pair(7)
    dev.to(device)
    x = normal("t/x", (2, 2, 257, 29), seed=1).to(device)
    target = uniform("t/y", (2, 1, 257, 29), 0.0, 1.0, seed=1).to(device)
    dev.eval()
    with torch.no_grad():
        y0 = dev(x)
    dev.train()
    opt = FlatAdam(dev.parameters(), lr=1e-2)
    l0 = train_step(dev, opt, x, target)
    l1 = train_step(dev, opt, x, target)
    assert l1.item() < l0.item()
    dev.eval()
    with torch.no_grad():
        y1 = dev(x)
    ref.load_state_dict({k: v.cpu() for k, v in dev.state_dict().items()}, strict=True)
    ref.eval()
    with torch.no_grad():
        want = ref(x.cpu())
    assert not torch.equal(y0, y1)
    assert_close(y1, want, what="eval forward after two optimizer steps")


def test_library_communicator_single_rank(device):
    """unetmi_comm_* / unetmi_allreduce_f32 (RCCL behind the C ABI) with a one-rank communicator: the sum
    all-reduce leaves the bucket as it was, FlatBucket.all_reduce takes this path once a communicator exists,
    and the entry points refuse misuse.  More than one rank needs more than one GPU (RCCL refuses two ranks on
    one device); the multi-rank plumbing around it is covered by the gloo tests on the host."""
    import unet_amd
    from unet_amd import _lib, comm
    from unet_amd.train import FlatBucket
    assert comm.world() == 0
    t = torch.arange(1000, dtype=torch.float32, device=device) * 0.5
    with pytest.raises(_lib.UnetmiError):
        comm.all_reduce_sum_(t.clone())                 # no communicator yet
    assert comm.init_from_torch_distributed(device) == 1
    try:
        assert comm.world() == 1
        with pytest.raises(_lib.UnetmiError):
            comm.init(0, 1, comm.unique_id(), device)   # one communicator per process
        got = comm.all_reduce_sum_(t.clone())
        torch.cuda.synchronize()
        assert torch.equal(got, t)
        p = torch.nn.Parameter(torch.ones(257, device=device))
        bucket = FlatBucket([p])
        bucket.grad.copy_(torch.linspace(-1, 1, 257, device=device))
        before = bucket.grad.clone()
        assert bucket.all_reduce() == 1
        torch.cuda.synchronize()
        assert torch.equal(bucket.grad, before)
    finally:
        comm.destroy()
    assert comm.world() == 0


def test_full_width_train_step_matches_oracle(device):
    """One training step of the full-width UNet (f_dim 32: the benchmark's channel counts, the 64- and 128-channel
    kernels and their weight-gradient instantiations) at B=2, T=45: loss, every gradient and the BatchNorm running
    statistics against the oracle.

    LeakyReLU's derivative jumps at zero.  The device's pre-activations differ from the CPU's in the last bits, so
    among 1.5 million of them per layer about one lands on the other side of zero, and that single element moves a
    per-channel gradient sum by up to 1e-2 of its value -- not an arithmetic error of the backward kernels.  The
    gradients are therefore compared with float64 autograd of the oracle run WITH THE DEVICE'S ACTIVATION PATTERN
    (the sign of every block output, captured from the device forward), to 1e-4 of each tensor's largest entry;
    the number of pre-activations whose sign differs from the oracle's own is reported and bounded."""
    import torch.nn as nn
    import oracle
    import unet_amd
    from unet_amd.seeding import normal, uniform
    from unet_amd.train import FlatAdam
    ref = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(ref, seed=31)
    dev = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    dev.load_state_dict(ref.state_dict(), strict=True)
    xb = normal("unet_full/xb", (2, 2, 257, 45), seed=32)
    target = uniform("unet_full/target", (2, 1, 257, 45), 0.0, 1.0, seed=32)
    ref64 = copy.deepcopy(ref).double().train()
    ref.train()
    loss_ref = torch.mean((ref(xb) - target) ** 2)

    dev.to(device).train()
    masks = {}
    handles = [m.register_forward_hook(lambda mod, i, o, n=name: masks.__setitem__(n, (o.detach() >= 0).cpu()))
               for name, m in dev.named_modules() if isinstance(m, (unet_amd.Encoder, unet_amd.Decoder))]
    opt = FlatAdam(dev.parameters(), lr=1e-3)
    opt.zero_grad()
    loss = torch.mean((dev(xb.to(device)) - target.to(device)) ** 2)
    loss.backward()
    for h in handles:
        h.remove()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    assert len(masks) == 20

    flips = [0, 0]

    def device_pattern(name):
        def hook(mod, inp, out):
            pre = inp[0]
            flips[0] += int(((pre >= 0) != masks[name]).sum())
            flips[1] += pre.numel()
            return pre * torch.where(masks[name], 1.0, mod.negative_slope).to(pre.dtype)
        return hook
    for name, m in ref64.named_modules():
        if isinstance(getattr(m, "acti", None), nn.LeakyReLU):
            m.acti.inplace = False          # the hook needs the pre-activation as it was
            m.acti.register_forward_hook(device_pattern(name))
    torch.mean((ref64(xb.double()) - target.double()) ** 2).backward()
    assert flips[1] > 1e6 and flips[0] <= 1e-5 * flips[1], flips
    exact = {k: p.grad for k, p in ref64.named_parameters()}
    gmax = max(g.abs().max().item() for g in exact.values())
    worst = 0.0
    for k, p in dev.named_parameters():
        g = p.grad.detach().cpu().double()
        scale = max(exact[k].abs().max().item(), 1e-3 * gmax)
        err = (g - exact[k]).abs().max().item()
        worst = max(worst, err / scale)
        assert err <= 1e-4 * scale, "grad {}: err {:.3e} of scale {:.3e}".format(k, err, scale)
    after_ref = ref.state_dict()
    for k, v in dev.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.detach().cpu(), after_ref[k], tol=1e-5, what="running statistic " + k)
    print("full-width train step: loss {:.6f}; {} of {} pre-activations on the other side of zero than the oracle's; "
          "worst gradient error {:.2e} of its tensor's scale".format(loss.item(), flips[0], flips[1], worst))


def test_dsunet_train_step_matches_oracle(device):
    """DSUNet in train() mode (DSUNet.py:582-628 under autograd): gated encoders, concat decoders, the 1x1 head and
    the tanh mask; loss on the complex mask's real and imaginary parts.  Every gradient -- including the attractor's
    Linears, which the gates feed -- against float64 autograd of the oracle run with the device's activation
    pattern (see test_full_width_train_step_matches_oracle), the BatchNorm running statistics against the fp32
    oracle."""
    import numpy as np
    import torch.nn as nn
    import oracle
    import unet_amd
    from unet_amd.DSUNet import Decoder as DSDecoder, Encoder as DSEncoder
    from unet_amd.seeding import normal, uniform
    ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8))
    unet_amd.seeded_fill_(ref, seed=51)
    dev = unet_amd.DSUNet(unet_amd.dsunet_architecture(45, 8))
    dev.load_state_dict(ref.state_dict(), strict=True)
    sf = normal("dsunet/sf_train", (2, 8, 257, 16), seed=52)
    SV = normal("dsunet/SV_train", (2, 4, 514), seed=52)
    ang = uniform("dsunet/theta_train", (2,), 0.0, 2 * np.pi, seed=52)
    theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
    target = uniform("dsunet/target_train", (2, 257, 16, 2), -1.0, 1.0, seed=52)

    def loss_of(net, *args, tgt):
        return torch.mean((torch.view_as_real(net(*args)) - tgt) ** 2)

    ref64 = copy.deepcopy(ref).double().train()
    ref.train()
    loss_ref = loss_of(ref, sf, SV, theta, tgt=target)

    dev.to(device).train()
    masks = {}
    handles = [m.register_forward_hook(lambda mod, i, o, n=name: masks.__setitem__(n, o.detach().cpu()))
               for name, m in dev.named_modules() if isinstance(m, (DSEncoder, DSDecoder))]
    loss = loss_of(dev, sf.to(device), SV.to(device), theta.to(device), tgt=target.to(device))
    loss.backward()
    for h in handles:
        h.remove()
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    assert len(masks) == 20

    flips = [0, 0]

    def device_pattern(name, block):
        def hook(mod, inp, out):
            pre = inp[0]
            # a block's device output is act(pre) times a positive gate (encoders): its sign is the pre-activation's
            pattern = masks[name] >= 0
            flips[0] += int(((pre >= 0) != pattern).sum())
            flips[1] += pre.numel()
            return pre * torch.where(pattern, 1.0, mod.negative_slope).to(pre.dtype)
        return hook
    for name, m in ref64.named_modules():
        if isinstance(getattr(m, "relu", None), nn.LeakyReLU):
            m.relu.inplace = False
            m.relu.register_forward_hook(device_pattern(name, m))
    loss_of(ref64, sf.double(), SV.double(), theta.double(), tgt=target.double()).backward()
    assert flips[1] > 1e6 and flips[0] <= 1e-5 * flips[1], flips
    exact = {k: p.grad for k, p in ref64.named_parameters()}
    gmax = max(g.abs().max().item() for g in exact.values() if g is not None)
    worst = 0.0
    report = []
    for k, p in dev.named_parameters():
        if exact[k] is None:                       # a parameter the forward does not use (in the reference as well)
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        assert p.grad is not None, k
        g = p.grad.detach().cpu().double()
        # the bias gradient of a convolution that feeds BatchNorm is exactly zero; what either side computes is
        # the rounding residue of a sum over ~1e5 terms, so the floor of the scale is 1e-2 of the largest gradient
        scale = max(exact[k].abs().max().item(), 1e-2 * gmax)
        err = (g - exact[k]).abs().max().item()
        worst = max(worst, err / scale)
        report.append((err / scale, k))
    if os.environ.get("UNETMI_TEST_VERBOSE") == "1":
        for e, k in sorted(report, reverse=True)[:12]:
            print("  {:50s} {:.2e}".format(k, e))
    bad = [(e, k) for e, k in report if e > 1e-4]
    assert not bad, bad[:5]
    after_ref = ref.state_dict()
    for k, v in dev.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.detach().cpu(), after_ref[k], tol=1e-5, what="running statistic " + k)
    print("DSUNet train step: loss {:.6f}; {} of {} pre-activations on the other side of zero; worst gradient error "
          "{:.2e} of its tensor's scale".format(loss.item(), flips[0], flips[1], worst))


def _gradients_against_float64(dev_net, ref, run, what, floor=1e-2, capture=None, tol=1e-4, cpu_factor=3.0):
    """Train-mode loss, gradients and BatchNorm running statistics of ``dev_net`` (on the device) against the oracle
    ``ref``.  ``run(net, cast)`` returns the scalar loss of a network on inputs converted by ``cast``.

    PReLU's and LeakyReLU's derivatives jump at zero, and the device's pre-activations differ from the CPU's in the
    last bits, so a handful among millions land on the other side of zero; each moves the gradient sums it feeds by
    up to 1e-2 of their value (the CPU's own fp32 autograd is that far from float64 autograd on these networks).
    The gradients are therefore compared with float64 autograd of the oracle run WITH THE DEVICE'S ACTIVATION
    PATTERN -- the sign of every pre-activation, captured call by call from the device forward -- to ``tol`` (1e-4) of
    each tensor's largest entry; the number of pre-activations whose sign differs from the oracle's own is reported and
    bounded.  ``capture(dev_net, patterns)`` records the patterns of activations that run inside a fused node on
    the device and that no module hook sees; it returns callables that undo its patching."""
    import unet_amd
    ref64 = copy.deepcopy(ref).double().train()
    ref.train()
    loss_ref = run(ref, lambda t: t)

    dev_net.train()
    patterns = {}

    def capture_input(name):
        return lambda mod, inp, out: patterns.setdefault(name, []).append((inp[0].detach() >= 0).cpu())

    def capture_positive(name):                                  # ReLU: the derivative is 1 for pre > 0
        return lambda mod, inp, out: patterns.setdefault(name, []).append((inp[0].detach() > 0).cpu())

    def capture_output(name):
        return lambda mod, inp, out: patterns.setdefault(name, []).append((out.detach() >= 0).cpu())
    handles = []
    for name, m in dev_net.named_modules():
        if isinstance(m, (nn.PReLU, nn.ReLU)):                   # runs as a torch module in train mode
            handles.append(m.register_forward_hook(capture_input(name) if isinstance(m, nn.PReLU) else capture_positive(name)))
        elif isinstance(m, (unet_amd.Encoder, unet_amd.Decoder)) and isinstance(m.acti, nn.LeakyReLU):
            handles.append(m.register_forward_hook(capture_output(name + ".acti")))   # inside the fused node: sign of the output
    undo = capture(dev_net, patterns) if capture is not None else []
    loss = run(dev_net, lambda t: t.to(next(dev_net.parameters()).device))
    loss.backward()
    for h in handles:
        h.remove()
    for u in undo:
        u()
    assert abs(loss.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item()), (loss.item(), loss_ref.item())

    flips, calls = [0, 0], {}

    def device_pattern(name, calls=calls, count=True):
        def hook(mod, inp, out):
            k = calls.get(name, 0)
            calls[name] = k + 1
            assert name in patterns, "no device pattern for " + name
            pre, mask = inp[0], patterns[name][k]
            if not count:
                slope = mod.weight if isinstance(mod, nn.PReLU) else torch.tensor(
                    0.0 if isinstance(mod, nn.ReLU) else mod.negative_slope, dtype=pre.dtype)
                return pre * torch.where(mask, torch.ones((), dtype=pre.dtype), slope)
            flips[0] += int((((pre > 0) if isinstance(mod, nn.ReLU) else (pre >= 0)) != mask).sum())
            flips[1] += pre.numel()
            if isinstance(mod, nn.PReLU):
                slope = mod.weight
            else:
                slope = torch.tensor(0.0 if isinstance(mod, nn.ReLU) else mod.negative_slope, dtype=pre.dtype)
            return pre * torch.where(mask, torch.ones((), dtype=pre.dtype), slope)
        return hook
    for name, m in ref64.named_modules():
        if isinstance(m, (nn.PReLU, nn.LeakyReLU, nn.ReLU)):
            if not isinstance(m, nn.PReLU):
                m.inplace = False              # the hook needs the pre-activation as it was
            m.register_forward_hook(device_pattern(name))
    run(ref64, lambda t: t.double()).backward()
    assert {k: len(v) for k, v in patterns.items()} == calls, "activation calls differ between device and oracle"
    assert flips[0] <= max(2, 2e-5 * flips[1]), flips
    exact = {k: p.grad for k, p in ref64.named_parameters()}
    # the CPU's own fp32 autograd on the same function (same pattern): how far fp32 rounding alone takes a gradient
    # from the float64 one -- where that exceeds the bar (normalisation over a handful of elements, a recurrence),
    # the device is held to ``cpu_factor`` times it
    ref32 = copy.deepcopy(ref).train()
    for p_ in ref32.parameters():
        p_.grad = None
    calls32 = {}
    for name, m in ref32.named_modules():
        if isinstance(m, (nn.PReLU, nn.LeakyReLU, nn.ReLU)):
            if not isinstance(m, nn.PReLU):
                m.inplace = False
            m.register_forward_hook(device_pattern(name, calls32, count=False))
    run(ref32, lambda t: t).backward()
    cpu32 = {k: p.grad for k, p in ref32.named_parameters()}
    gmax = max(g.abs().max().item() for g in exact.values() if g is not None)
    worst = worst_cpu = 0.0
    report, failures = [], []
    for k, p in dev_net.named_parameters():
        if exact[k] is None:
            assert p.grad is None or p.grad.abs().max().item() == 0.0, k
            continue
        assert p.grad is not None, "no gradient for " + k
        g = p.grad.detach().cpu().double()
        scale = max(exact[k].abs().max().item(), floor * gmax)
        err = (g - exact[k]).abs().max().item()
        cpu_err = (cpu32[k].double() - exact[k]).abs().max().item()
        worst, worst_cpu = max(worst, err / scale), max(worst_cpu, cpu_err / scale)
        report.append((err / scale, cpu_err / scale, k))
        if err > max(tol * scale, cpu_factor * cpu_err):
            failures.append("grad {}: err {:.3e}, CPU fp32 err {:.3e}, scale {:.3e}".format(k, err, cpu_err, scale))
    if os.environ.get("UNETMI_TEST_VERBOSE") == "1":
        for e, c, k in sorted(report, reverse=True)[:25]:
            print("  {:50s} device {:.2e}  CPU fp32 {:.2e}".format(k, e, c))
    assert not failures, failures[:5]
    after_ref = ref.state_dict()
    for k, v in dev_net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert_close(v.detach().cpu(), after_ref[k], tol=1e-5, what="running statistic " + k)
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(after_ref[k]), k
    print("{}: loss {:.6f}; {} of {} pre-activations on the other side of zero than the oracle's; worst gradient "
          "error {:.2e} of its tensor's scale (CPU fp32 autograd with the same pattern: {:.2e})".format(
              what, loss.item(), flips[0], flips[1], worst, worst_cpu))


@pytest.mark.parametrize("kernel,dilation", [((3, 3), (1, 1)), ((3, 3), (1, 4)), ((1, 3), (1, 2)), ((3, 1), (2, 1))])
def test_depthwise_conv_gradients(device, kernel, dilation):
    """Autograd of unetmi::dwconv2d_fwd (the depthwise convolution of a ResBlock layer, UNet_m.py:546-548): input,
    addend, weight and bias gradients against float64 autograd of torch's grouped convolution."""
    import torch.nn.functional as F
    from unet_amd.ops import dwconv2d_fwd
    C = 12
    pad = [dilation[0] * (kernel[0] - 1) // 2, dilation[1] * (kernel[1] - 1) // 2]
    x, x2 = _rand((3, C, 9, 37), 5), _rand((3, C, 9, 37), 6)
    w, b, g = _rand((C, 1) + kernel, 7, 0.5), _rand((C,), 8), _rand((3, C, 9, 37), 9)
    leaves64 = [t.double().requires_grad_() for t in (x, x2, w, b)]
    F.conv2d(leaves64[0] + leaves64[1], leaves64[2], leaves64[3], 1, pad, dilation, C).backward(g.double())
    leaves = [t.to(device).requires_grad_() for t in (x, x2, w, b)]
    out = dwconv2d_fwd(leaves[0], leaves[1], leaves[2], leaves[3], pad, list(dilation))
    out.backward(g.to(device))
    for name, a, e in zip(("x", "x2", "weight", "bias"), leaves, leaves64):
        assert_close(a.grad.cpu(), e.grad.float(), tol=2e-5, what="depthwise conv gradient of " + name)


@pytest.mark.parametrize("channels", [64, 30])
def test_resblock_train_step_matches_oracle(device, channels):
    """ResBlock in train() mode (UNet_m.py:518-563 under autograd): five layers of depthwise conv, pointwise conv,
    the shared PReLU and BatchNorm with batch statistics, at the width of ResUNet (64) and of ResUNetOnFreq (30)."""
    import oracle
    import unet_amd
    from unet_amd.seeding import normal
    ref = oracle.ResBlock(channels)
    unet_amd.seeded_fill_(ref, seed=61)
    dev_net = unet_amd.ResBlock(channels)
    dev_net.load_state_dict(ref.state_dict(), strict=True)
    dev_net.to(device)
    x = normal("resblock_train/x", (2, channels, 17, 40), seed=62)
    target = normal("resblock_train/target", (2, channels, 17, 40), seed=63)
    _gradients_against_float64(dev_net, ref, lambda net, cast: torch.mean((net(cast(x)) - cast(target)) ** 2),
                               "ResBlock({}) train step".format(channels))


@pytest.mark.parametrize("B,T,layers,dims", [(5, 12, 3, (384, 640)), (32, 37, 2, (384, 640)), (37, 9, 1, (384, 640)),
                                             (4, 501, 3, (384, 640)), (6, 21, 3, (192, 300)), (32, 14, 2, (210, 300))])
def test_projected_lstm_gradients(device, B, T, layers, dims):
    """The projected LSTM as an autograd node (ResUNet.py:139: nn.LSTM(384, 640, 3, proj_size=384, batch_first);
    ResUNet.py:335: LSTM(192, 300, 3, proj_size=192); UNet_m.py:587-610: LSTM(210, 300, 3, proj_size=210)):
    forward recurrence saving gates and cell states, backpropagation through time as one persistent launch per layer,
    weight gradients as time-batched products.  Output, input gradient and all 5 parameter gradients per layer
    against float64 autograd of torch's CPU LSTM; B=37 runs as two groups of sequences, T=501 is the benchmark's
    sequence length."""
    from unet_amd import lstm as L
    torch.manual_seed(7)
    C, H = dims
    ref = nn.LSTM(C, H, layers, batch_first=True, proj_size=C)
    dev_lstm = copy.deepcopy(ref).to(device)
    ref = ref.double()
    x = _rand((B, T, C), 21)
    g = _rand((B, T, C), 22)
    if not L.trainable_on_device(dev_lstm, x.to(device)):
        pytest.skip("the XCD-local recurrence kernels need a device of 8 XCDs x 32 compute units")
    x64 = x.double().requires_grad_()
    y64 = ref(x64)[0]
    y64.backward(g.double())
    xd = x.to(device).requires_grad_()
    y = L.lstm_forward_train(dev_lstm, xd)
    assert_close(y.detach().cpu(), y64.detach().float(), tol=2e-5, what="projected LSTM output")
    y.backward(g.to(device), retain_graph=True)
    assert_close(xd.grad.cpu(), x64.grad.float(), tol=2e-5, what="projected LSTM input gradient")
    exact = dict(ref.named_parameters())
    for k, p in dev_lstm.named_parameters():
        assert p.grad is not None, k
        assert_close(p.grad.cpu(), exact[k].grad.float(), tol=2e-5, what="projected LSTM gradient of " + k)
    # a second backward pass through the same graph would read gates that are gradients by now: refused, not wrong
    with pytest.raises(RuntimeError, match="retain_graph"):
        y.backward(g.to(device))
    # a second step on the same module: nothing from the first call is kept
    for p in dev_lstm.parameters():
        p.grad = None
    y2 = L.lstm_forward_train(dev_lstm, xd.detach())
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("C,H,B,T,with_h0", [(128, 128, 5, 19, False), (128, 128, 48, 33, True), (256, 512, 3, 11, True),
                                               (256, 256, 70, 7, False), (128, 128, 2, 512, False)])
def test_gru_gradients(device, C, H, B, T, with_h0):
    """nn.GRU as an autograd node on the persistent kernels (DSUNet.py:281-312 TGRUBlock GRU(128,128); UDSS.py:39-70
    GRU(256,512); UDSS.py:190-239 GRU(256,256)): outputs, final state, input / initial-state gradients and the four
    parameter gradients against float64 autograd of torch's CPU GRU.  The loss uses the outputs AND the final state;
    B=70 runs as two groups of sequences, T=512 is DSUNet's full sequence length."""
    from unet_amd import gru as G
    torch.manual_seed(11)
    ref = nn.GRU(C, H, 1, batch_first=True)
    dev_gru = copy.deepcopy(ref).to(device).train()
    ref = ref.double()
    x, g, gh = _rand((B, T, C), 31), _rand((B, T, H), 32), _rand((1, B, H), 33)
    h0 = _rand((1, B, H), 34, 0.5) if with_h0 else None
    x64 = x.double().requires_grad_()
    h64 = h0.double().requires_grad_() if with_h0 else None
    y64, hn64 = ref(x64, h64)
    ((y64 * g.double()).sum() + (hn64 * gh.double()).sum()).backward()
    xd = x.to(device).requires_grad_()
    hd = h0.to(device).requires_grad_() if with_h0 else None
    y, hn = G.run(dev_gru, xd, hd)
    assert y.grad_fn is not None and "Gru" in type(y.grad_fn).__name__ or B > 64, type(y.grad_fn)
    assert_close(y.detach().cpu(), y64.detach().float(), tol=2e-5, what="GRU output")
    assert_close(hn.detach().cpu(), hn64.detach().float(), tol=2e-5, what="GRU final state")
    ((y * g.to(device)).sum() + (hn * gh.to(device)).sum()).backward()
    assert_close(xd.grad.cpu(), x64.grad.float(), tol=2e-5, what="GRU input gradient")
    if with_h0:
        assert_close(hd.grad.cpu(), h64.grad.float(), tol=2e-5, what="GRU initial-state gradient")
    exact = dict(ref.named_parameters())
    for k, p in dev_gru.named_parameters():
        assert p.grad is not None, k
        assert_close(p.grad.cpu(), exact[k].grad.float(), tol=2e-5, what="GRU gradient of " + k)


def test_resunet_train_step_matches_oracle(device):
    """ResUNet in train() mode (ResUNet.py:161-220 under autograd) at B=2, T=16: input layers, ResBlocks, PReLU
    encoders / decoders, 1x1 res paths, the projected LSTM (unet_amd.lstm.lstm_forward_train), additive skips, head
    and mask.  Loss, every gradient and the running statistics against the oracle."""
    import oracle
    import unet_amd
    from unet_amd.seeding import normal
    ref = oracle.ResUNet(c_in=4, c_out=1)
    unet_amd.seeded_fill_(ref, seed=71)
    dev_net = unet_amd.ResUNet(c_in=4, c_out=1)
    dev_net.load_state_dict(ref.state_dict(), strict=True)
    dev_net.to(device)
    x = normal("resunet_train/x", (2, 4, 257, 16), seed=72)
    target = normal("resunet_train/target", (2, 2, 257, 16), seed=73)
    _gradients_against_float64(dev_net, ref, lambda net, cast: torch.mean((net(cast(x)) - cast(target)) ** 2),
                               "ResUNet train step")


@pytest.mark.parametrize("variant", ["ResUNetOnFreq", "ResUNetOnFreq2"])
def test_resunet_on_freq_train_step_matches_oracle(device, variant):
    """The 30-channel variants in train() mode (ResUNet.py:227-389, :394-591 under autograd), Softplus mask;
    ResUNetOnFreq2 includes the reference's ``self.bottleneck(x)[0]`` (sample 0's bottleneck output for the whole
    batch), whose gradient autograd sums back onto sample 0."""
    import oracle
    import unet_amd
    from unet_amd.seeding import normal
    c_in = 4 if variant == "ResUNetOnFreq" else 1
    ref = getattr(oracle, variant)(c_in=c_in, c_out=1)
    unet_amd.seeded_fill_(ref, seed=81)
    dev_net = getattr(unet_amd, variant)(c_in=c_in, c_out=1)
    dev_net.load_state_dict(ref.state_dict(), strict=True)
    dev_net.to(device)
    x = normal("onfreq_train/x", (2, c_in, 257, 12), seed=82)
    target = normal("onfreq_train/target", (2, 1, 257, 12), seed=83)
    _gradients_against_float64(dev_net, ref, lambda net, cast: torch.mean((net(cast(x)) - cast(target)) ** 2),
                               variant + " train step")


@pytest.mark.parametrize("family", ["UDSS", "DAU"])
def test_complex_family_train_step_matches_oracle(device, family):
    """UDSS / DAU in train() mode (UDSS.py:561-580, DTU.py:340-378 under autograd): gated complex encoders, complex
    decoders over concatenated skips, the complex 1x1 head and the tanh mask.  Every gradient -- the real and the
    imaginary convolution of each complex layer, both BatchNorms, the attractor's Linears -- against float64
    autograd of the oracle with the device's activation pattern."""
    import oracle
    import unet_amd
    from unet_amd.UDSS import Decoder as CDecoder, Encoder as CEncoder, unstack_complex
    from unet_amd.seeding import normal, uniform
    kw = dict(type_norm="BatchNorm2d") if family == "UDSS" else {}
    ref = getattr(oracle, family)(**kw)
    unet_amd.seeded_fill_(ref, seed=91)
    dev_net = getattr(unet_amd, family)(**kw)
    dev_net.load_state_dict(ref.state_dict(), strict=True)
    dev_net.to(device)
    sf = normal("udss_train/sf", (2, 4, 257, 16, 2), seed=92)
    SV = normal("udss_train/SV", (2, 4, 514), seed=92)
    ang = uniform("udss_train/theta", (2,), 0.0, 2 * np.pi, seed=92)
    theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
    target = uniform("udss_train/target", (2, 257, 16, 2), -1.0, 1.0, seed=92)

    def capture(net, patterns):
        # the blocks are called through forward_stacked, and their LeakyReLU runs inside the fused node: a block's
        # output is act(pre) times a positive gate, so its sign is the pre-activation's
        undo = []
        for name, m in net.named_modules():
            if isinstance(m, (CEncoder, CDecoder)):
                inner = m.forward_stacked

                def wrapped(*a, _inner=inner, _name=name, **k):
                    out = _inner(*a, **k)
                    patterns.setdefault(_name + ".relu", []).append((unstack_complex(out.detach()) >= 0).cpu())
                    return out
                m.forward_stacked = wrapped
                undo.append(lambda _m=m: _m.__dict__.pop("forward_stacked"))
        return undo

    def run(net, cast):
        return torch.mean((torch.view_as_real(net(cast(sf), cast(SV), cast(theta))) - cast(target)) ** 2)
    # every tensor but one stays below 8e-5; the BatchNorm weight of one deep encoder (statistics over 48 elements
    # per channel at this size, a gradient 100 times smaller than the largest) reaches 1.2e-4, so the bar is 2e-4
    _gradients_against_float64(dev_net, ref, run, family + " train step", capture=capture, tol=2e-4)


@pytest.mark.parametrize("variant,kw", [("gconv", dict(type_encoder="GConv")), ("tgru", dict(bottleneck="TGRU")),
                                        ("prelu_instnorm", dict(activation="PReLU", type_norm="InstanceNorm2d"))])
def test_dsunet_variants_train_step_matches_oracle(device, variant, kw):
    """DSUNet's other block types in train() mode: gated convolutions (the two convolutions as one differentiable
    node on the stacked weight), the time-GRU bottleneck (torch's device GRU under autograd, then conv + BatchNorm +
    ReLU as one node), and PReLU / InstanceNorm2d blocks (torch operations around the convolution node)."""
    import oracle
    import unet_amd
    from unet_amd.DSUNet import Decoder as DSDecoder, Encoder as DSEncoder, TGRUBlock
    from unet_amd.seeding import normal, uniform
    ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8), **kw)
    unet_amd.seeded_fill_(ref, seed=53)
    dev_net = unet_amd.DSUNet(unet_amd.dsunet_architecture(45, 8), **kw)
    dev_net.load_state_dict(ref.state_dict(), strict=True)
    dev_net.to(device)
    sf = normal("dsunet/sf_train", (2, 8, 257, 16), seed=54)
    SV = normal("dsunet/SV_train", (2, 4, 514), seed=54)
    ang = uniform("dsunet/theta_train", (2,), 0.0, 2 * np.pi, seed=54)
    theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
    target = uniform("dsunet/target_train", (2, 257, 16, 2), -1.0, 1.0, seed=54)

    def capture(net, patterns):
        # activations that run inside a fused node: the sign of the block's output is the pre-activation's
        # (LeakyReLU times a positive gate; ReLU: positive output <=> positive pre-activation)
        handles = []
        for name, m in net.named_modules():
            if isinstance(m, (DSEncoder, DSDecoder)) and isinstance(m.relu, nn.LeakyReLU) and isinstance(m.bn, nn.BatchNorm2d):
                handles.append(m.register_forward_hook(
                    lambda mod, i, o, n=name: patterns.setdefault(n + ".relu", []).append((o.detach() >= 0).cpu())))
            elif isinstance(m, TGRUBlock):
                handles.append(m.register_forward_hook(
                    lambda mod, i, o, n=name: patterns.__setitem__(n + ".relu", [(o.detach() > 0).cpu()])))
        return [h.remove for h in handles]

    def run(net, cast):
        return torch.mean((torch.view_as_real(net(cast(sf), cast(SV), cast(theta))) - cast(target)) ** 2)
    _gradients_against_float64(dev_net, ref, run, "DSUNet {} train step".format(variant), capture=capture)
