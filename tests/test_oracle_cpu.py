"""The oracle against the committed golden vectors (which were produced by the
reference itself, tests/golden/make_golden.py).  Runs without a GPU."""

import numpy as np
import pytest
import torch

import oracle
from tests.util import assert_close, load_golden, sample_index
from unet_amd.seeding import normal, seeded_fill_, uniform

CPU_TOL = 2e-5   # same arithmetic, possibly a different CPU ISA path than the build container


def _sampled(t, n=2048):
    if t.is_complex():
        t = torch.view_as_real(t)
    flat = t.detach().reshape(-1)
    return flat[torch.from_numpy(sample_index(flat.numel(), n))]


def test_unet_narrow_matches_golden():
    net = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=8)).eval()
    seeded_fill_(net, seed=11)
    x = normal("unet_narrow/x", (1, 2, 257, 29), seed=11)
    g = load_golden("unet_narrow_eval")
    store = {}
    names = ["encoder_{}".format(i) for i in range(10)] + ["decoder_{}".format(i) for i in range(10)]
    mods = dict(net.named_modules())
    hooks = [mods[n].register_forward_hook(lambda m, i, o, n=n: store.__setitem__(n, o.double()))
             for n in names]
    with torch.no_grad():
        y = net(x)
    for h in hooks:
        h.remove()
    assert_close(y, torch.from_numpy(g["out"]), tol=CPU_TOL, what="UNet narrow")
    got = np.array([store[n].sum().item() for n in names])
    np.testing.assert_allclose(got, g["layer_sum"], rtol=1e-4, atol=1e-3)


def test_unet_benchmark_size_matches_golden():
    """Full channel widths at T=509 (one utterance): output summary recorded from the reference."""
    m = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32)).eval()
    seeded_fill_(m, seed=13)
    x = normal("unet_full509/x", (1, 2, 257, 509), seed=13)
    with torch.no_grad():
        y = m(x).reshape(-1).double().numpy()
    g = load_golden("unet_full_eval_T509")
    assert y.size == int(g["out_numel"])
    assert np.abs(y[sample_index(y.size)] - g["out_samples"]).max() <= CPU_TOL * np.abs(g["out_samples"]).max()
    assert abs(y.sum() - float(g["out_sum"])) <= CPU_TOL * y.size


@pytest.mark.parametrize("T", [29, 125])
def test_unet20_matches_golden(T):
    net = oracle.UNet20().eval()
    seeded_fill_(net, seed=21)
    net.T = T
    x = uniform("unet20/x{}".format(T), (1, 14, 257, T), 0.0, 1.0, seed=21)
    with torch.no_grad():
        y = net(x)
    assert tuple(y.shape) == (1, 4, 4, 1, 1, 257, T) and y.dtype == torch.complex64
    g = load_golden("unet20_eval_T{}".format(T))
    assert_close(_sampled(y), torch.from_numpy(g["out_samples"]), tol=CPU_TOL, what="UNet20 samples")
    assert abs(torch.view_as_real(y).double().sum().item() - float(g["out_sum"])) <= 1e-5 * abs(float(g["out_sum"]))


def test_unet20_complex_pairs_adjacent_time_frames():
    """The reference reshapes the NCHW buffer raw: re/im of one complex value are
    two ADJACENT TIME FRAMES of one channel (cRFUNet.py:320-324)."""
    net = oracle.UNet20().eval()
    seeded_fill_(net, seed=21)
    net.T = 29
    x = uniform("unet20/x29", (1, 14, 257, 29), 0.0, 1.0, seed=21)
    with torch.no_grad():
        y = net(x)
        p = oracle.nets._concat_shell_forward(net, x)
        f = net.activation_mask(net.linear(p))
    flat = f.reshape(-1)
    first = y.reshape(-1)[0]
    assert first.real.item() == flat[0].item() and first.imag.item() == flat[1].item()
    assert flat[1].item() == f[0, 0, 0, 1].item()


def test_resunet_and_resblock_match_golden():
    net = oracle.ResUNet().eval()
    seeded_fill_(net, seed=31)
    x = normal("resunet/x", (1, 4, 257, 16), seed=31)
    with torch.no_grad():
        y = net(x)
    assert_close(y, torch.from_numpy(load_golden("resunet_eval")["out"]), tol=CPU_TOL, what="ResUNet")
    rb = oracle.ResBlock(16).eval()
    seeded_fill_(rb, seed=32)
    with torch.no_grad():
        y = rb(normal("resblock/x", (2, 16, 9, 40), seed=32))
    assert_close(y, torch.from_numpy(load_golden("resblock_eval")["out"]), tol=CPU_TOL, what="ResBlock")


def test_resunet_onfreq_matches_golden():
    for label, kw in (("softplus", {}), ("sigmoid", dict(activation="Sigmoid", c_in=6))):
        net = oracle.ResUNetOnFreq(**kw).eval()
        seeded_fill_(net, seed=33)
        x = normal("resunet_onfreq/x_" + label, (2, kw.get("c_in", 4), 257, 16), seed=33)
        with torch.no_grad():
            y = net(x)
        assert tuple(y.shape) == (2, 1, 257, 16)
        assert_close(y, torch.from_numpy(load_golden("resunet_onfreq_eval_" + label)["out"]), tol=CPU_TOL,
                     what="ResUNetOnFreq " + label)


def test_resunet_onfreq2_matches_golden():
    for label, kw in (("softplus", {}), ("tanh", dict(activation="Tanh", c_in=2)), ("ms", dict(multi_scale=True)),
                      ("mea", dict(c_in=2, c_out=2, activation="MEA", multi_scale=True))):
        net = oracle.ResUNetOnFreq2(**kw).eval()
        seeded_fill_(net, seed=34)
        x = normal("resunet_onfreq2/x_" + label, (2, kw.get("c_in", 1), 257, 16), seed=34)
        with torch.no_grad():
            y = net(x)
        if label == "mea":                                   # the MEA head: three masks
            gold = load_golden("resunet_onfreq2_eval_mea")
            for name, t in zip(("mag_mask", "real_mask", "imag_mask"), y):
                assert_close(t, torch.from_numpy(gold[name]), tol=CPU_TOL, what="ResUNetOnFreq2 MEA " + name)
            with pytest.raises(AttributeError):              # ResUNet.py:589 names an attribute never set
                net.output(y, x)
            continue
        assert_close(y, torch.from_numpy(load_golden("resunet_onfreq2_eval_" + label)["out"]), tol=CPU_TOL,
                     what="ResUNetOnFreq2 " + label)


def test_udss_matches_golden():
    gold = load_golden("udss_eval")
    for label, kw in (("crm", {}), ("cm", dict(type_masking="CM")), ("tgru", dict(bottleneck="TGRU")),
                      ("ftgru", dict(bottleneck="FTGRU")), ("crnn", dict(bottleneck="cRNN"))):
        net = oracle.UDSS(type_norm="BatchNorm2d", **kw).eval()
        seeded_fill_(net, seed=61)
        sf = normal("udss/sf", (2, 4, 257, 16, 2), seed=61)
        SV = normal("udss/SV", (2, 4, 514), seed=61)
        ang = uniform("udss/theta", (2,), 0.0, 2 * np.pi, seed=61)
        theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
        with torch.no_grad():
            y = net(sf, SV, theta)
        assert y.dtype == torch.complex64 and tuple(y.shape) == (2, 257, 16)
        assert_close(torch.view_as_real(y), torch.from_numpy(gold[label]), tol=CPU_TOL, what="UDSS " + label)
    net = oracle.UDSS(type_norm="BatchNorm2d", type_masking="CMEA").eval()
    seeded_fill_(net, seed=61)
    X = torch.complex(normal("udss/Xr", (2, 257, 16), seed=62), normal("udss/Xi", (2, 257, 16), seed=62))
    with torch.no_grad():
        masks = net(sf, SV, theta)
        applied = net.masking(X, masks)
    assert_close(torch.stack(masks), torch.from_numpy(gold["cmea_masks"]), tol=CPU_TOL, what="UDSS CMEA masks")
    assert_close(applied, torch.from_numpy(gold["cmea_applied"]), tol=CPU_TOL, what="UDSS CMEA applied")
    # the reference's default type_norm is a misspelt name that leaves the norm class unbound
    try:
        oracle.UDSS()
        raise AssertionError("UDSS() should fail like the reference")
    except KeyError:
        pass


def test_udss_helper_matches_golden():
    gold = load_golden("udss_helper_eval")
    for label, kw in (("plain", {}), ("dsb", dict(DSB=True))):
        net = oracle.UDSS_helper(**kw).eval()
        seeded_fill_(net, seed=66)
        wav = normal("udss_helper/x", (2, 4, 4000), seed=66) * 0.1
        angle = uniform("udss_helper/angle", (2,), 0.0, 360.0, seed=66)
        mic = uniform("udss_helper/mic", (2, 4, 3), -0.05, 0.05, seed=66)
        with torch.no_grad():
            y = net(wav, angle, mic)
        assert tuple(y.shape) == (2, 4000)
        assert_close(y, torch.from_numpy(gold[label]), tol=CPU_TOL, what="UDSS_helper " + label)


def test_dau_matches_golden():
    """DTU.py's DAU and DAU_helper restatements against the reference's outputs (tests/golden/make_golden.py)."""
    gold = load_golden("dau_eval")
    for label, kw in (("crm", {}), ("mea", dict(type_masking="MEA"))):
        net = oracle.DAU(**kw).eval()
        seeded_fill_(net, seed=81)
        sf = normal("dau/sf", (2, 4, 257, 16, 2), seed=81)
        SV = normal("dau/SV", (2, 4, 514), seed=81)
        ang = uniform("dau/theta", (2,), 0.0, 2 * np.pi, seed=81)
        theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
        with torch.no_grad():
            y = net(sf, SV, theta)
        assert y.dtype == torch.complex64 and tuple(y.shape) == (2, 257, 16)
        assert_close(torch.view_as_real(y), torch.from_numpy(gold[label]), tol=CPU_TOL, what="DAU " + label)
    gold = load_golden("dau_helper_eval")
    for label, kw in (("plain", {}), ("dsb", dict(DSB=True))):
        net = oracle.DAU_helper(**kw).eval()
        seeded_fill_(net, seed=86)
        wav = normal("dau_helper/x", (2, 4, 4000), seed=86) * 0.1
        angle = uniform("dau_helper/angle", (2,), 0.0, 360.0, seed=86)
        mic = uniform("dau_helper/mic", (2, 4, 3), -0.05, 0.05, seed=86)
        with torch.no_grad():
            y = net(wav, angle, mic)
        assert_close(y, torch.from_numpy(gold[label]), tol=CPU_TOL, what="DAU_helper " + label)
    # the reference's constructor refuses these (Exception / TypeError from nn.PReLU(inplace=True))
    for bad, exc in ((dict(bottleneck="TGRU"), Exception), (dict(activation="PReLU"), TypeError),
                     (dict(type_masking="CM"), Exception)):
        try:
            oracle.DAU(**bad)
            raise AssertionError("DAU({}) should fail like the reference".format(bad))
        except exc:
            pass


def test_dsunet_matches_golden():
    net = oracle.DSUNet(oracle.dsunet_architecture(45, 8)).eval()
    seeded_fill_(net, seed=41)
    sf = normal("dsunet/sf", (2, 8, 257, 16), seed=41)
    SV = normal("dsunet/SV", (2, 4, 514), seed=41)
    ang = uniform("dsunet/theta", (2,), 0.0, 2 * np.pi, seed=41)
    theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
    with torch.no_grad():
        M = net(sf, SV, theta)
    g = load_golden("dsunet_eval")
    assert_close(torch.view_as_real(M), torch.from_numpy(g["out"]), tol=CPU_TOL, what="DSUNet")
    X = torch.complex(normal("dsunet/Xr", (2, 257, 16), seed=41), normal("dsunet/Xi", (2, 257, 16), seed=41))
    assert_close(torch.view_as_real(net.masking(X, M)), torch.from_numpy(g["masked"]), tol=CPU_TOL, what="CRM")


def test_unet_train_step_matches_golden():
    """fwd (batch statistics), MSE, backward, one Adam step on the narrow UNet."""
    net = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=8))
    seeded_fill_(net, seed=11)
    net.train()
    xb = normal("unet_narrow/xb", (2, 2, 257, 29), seed=12)
    target = uniform("unet_narrow/target", (2, 1, 257, 29), 0.0, 1.0, seed=12)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    loss = torch.mean((net(xb) - target) ** 2)
    loss.backward()
    g = load_golden("unet_narrow_trainstep")
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    for k, p in net.named_parameters():
        want = g["gradsum/" + k]
        assert abs(p.grad.double().sum().item() - float(want)) <= 1e-3 * (abs(float(want)) + 1e-6), k
    opt.step()
    sd = net.state_dict()
    for k in ("encoder_2.bn.running_mean", "decoder_7.bn.running_var", "linear.weight"):
        got = sd[k].float().reshape(-1)[torch.from_numpy(sample_index(sd[k].numel(), 8))]
        np.testing.assert_allclose(got.numpy(), g["post/" + k], rtol=2e-4, atol=1e-6)


def test_crf_apply_degenerates_to_product():
    """With 1x1 taps the (unpinned) ratio-filter application is H * X."""
    H = torch.complex(torch.randn(2, 4, 4, 1, 1, 9, 11), torch.randn(2, 4, 4, 1, 1, 9, 11))
    X = torch.complex(torch.randn(2, 4, 9, 11), torch.randn(2, 4, 9, 11))
    Y = oracle.crf_apply(H, X)
    assert torch.allclose(Y, H[:, :, :, 0, 0] * X[:, None])


def test_state_dict_keys_and_shape_rules():
    """142 keys for the UNet shell; the table only accepts T = 13 (mod 16)."""
    net = oracle.UNet(*oracle.default_unet(f_dim=8)).eval()
    assert len(net.state_dict()) == 142
    with torch.no_grad():
        assert tuple(net(torch.zeros(1, 2, 257, 13)).shape) == (1, 1, 257, 13)
        with pytest.raises(RuntimeError):
            net(torch.zeros(1, 2, 257, 16))
    with pytest.raises(Exception, match="unmatched legnth"):
        oracle.UNet([oracle.Encoder(2, 4, (3, 3), 1)], [])
    with pytest.raises(Exception, match="Unknown activation"):
        oracle.Encoder(2, 4, (3, 3), 1, activation="Swish")


def test_dsunet_helper_matches_golden():
    """Wave-in / wave-out wrapper (STFT, steering-vector features, DSUNet, mask, inverse STFT)."""
    arch = oracle.dsunet_architecture(45, 8)
    m = oracle.DSUNet_helper(arch).eval()
    seeded_fill_(m, seed=51)
    wav = normal("helper/x", (2, 4, 4000), seed=51) * 0.1
    angle = uniform("helper/angle", (2,), 0.0, 360.0, seed=51)
    mic = uniform("helper/mic", (2, 4, 3), -0.05, 0.05, seed=51)
    with torch.no_grad():
        y = m(wav, angle, mic)
    assert_close(y, torch.from_numpy(load_golden("dsunet_helper_eval")["out"]), tol=CPU_TOL, what="DSUNet_helper")


@pytest.mark.parametrize("variant,kw", [("gconv", dict(type_encoder="GConv")), ("tgru", dict(bottleneck="TGRU")),
                                        ("crm_soft", dict(type_masking="CRM_soft")), ("cm", dict(type_masking="CM"))])
def test_dsunet_variants_match_golden(variant, kw):
    m = oracle.DSUNet(oracle.dsunet_architecture(45, 8), **kw).eval()
    seeded_fill_(m, seed=43)
    sf = normal("dsunet/sf", (2, 8, 257, 16), seed=41)
    SV = normal("dsunet/SV", (2, 4, 514), seed=41)
    ang = uniform("dsunet/theta", (2,), 0.0, 2 * np.pi, seed=41)
    theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
    with torch.no_grad():
        y = m(sf, SV, theta)
    golden = torch.view_as_complex(torch.from_numpy(load_golden("dsunet_variants_eval")[variant]).contiguous())
    assert_close(y, golden, tol=CPU_TOL, what="DSUNet " + variant)
