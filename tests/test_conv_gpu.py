"""Parity of the fused HIP convolution (through the C ABI / torch.ops.unetmi)
against the torch CPU ops the reference calls at the same sites, on every
(kernel, stride, padding, output_padding, dilation) tuple of the layer tables
(cRFUNet.py:192-218, ResUNet.py:47-142, UDSS.py:595-683) plus ragged and
degenerate sizes."""

import pytest
import torch
import torch.nn as nn

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _fill(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (0.5 if p.dim() > 1 else 0.3))
        for name, b in mod.named_buffers():
            if name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                b.copy_(torch.rand(b.shape, generator=g) - 0.5)


# (transposed, Cin, Cout, k, s, p, op, d, H, W)
GEOMS = [
    # T1 encoders
    (0, 2, 32, (7, 1), (1, 1), (3, 0), (0, 0), (1, 1), 37, 45),
    (0, 14, 32, (7, 1), (1, 1), (3, 0), (0, 0), (1, 1), 20, 33),
    (0, 8, 64, (1, 7), (1, 1), (0, 3), (0, 0), (1, 1), 19, 70),
    (0, 6, 64, (7, 5), (2, 2), (3, 2), (0, 0), (1, 1), 41, 77),
    (0, 6, 64, (7, 5), (2, 1), (3, 2), (0, 0), (1, 1), 33, 40),
    (0, 10, 16, (5, 3), (2, 2), (2, 1), (0, 0), (1, 1), 33, 31),
    (0, 10, 70, (5, 3), (2, 1), (2, 1), (0, 0), (1, 1), 9, 64),
    (0, 4, 8, (5, 3), (2, 1), (2, 1), (0, 0), (1, 1), 3, 32),
    # T1 decoders
    (1, 8, 16, (5, 3), (2, 1), (2, 1), (0, 0), (1, 1), 2, 32),
    (1, 12, 16, (5, 3), (2, 2), (2, 1), (0, 1), (1, 1), 3, 32),
    (1, 12, 64, (5, 3), (2, 2), (2, 1), (0, 0), (1, 1), 17, 37),
    (1, 6, 64, (7, 5), (2, 1), (3, 2), (0, 0), (1, 1), 21, 45),
    (1, 6, 64, (7, 5), (2, 2), (3, 2), (0, 0), (1, 1), 23, 39),
    (1, 10, 33, (1, 7), (1, 1), (0, 3), (0, 0), (1, 1), 11, 50),
    (1, 12, 1, (7, 1), (1, 1), (3, 0), (0, 0), (1, 1), 30, 40),
    (0, 1, 96, (7, 1), (1, 1), (3, 0), (0, 0), (1, 1), 21, 70),      # the input gradient of dec_9 as a layer
    (1, 2, 40, (3, 1), (1, 1), (1, 0), (0, 0), (1, 1), 9, 33),
    # heads / res paths
    (0, 1, 1, (1, 1), (1, 1), (0, 0), (0, 0), (1, 1), 13, 29),
    (0, 16, 32, (1, 1), (1, 1), (0, 0), (0, 0), (1, 1), 13, 29),
    # T3 (ResUNet)
    (0, 2, 20, (1, 3), (1, 1), (0, 1), (0, 0), (1, 1), 16, 257),
    (0, 8, 8, (3, 3), (2, 1), (0, 1), (0, 0), (1, 1), 31, 20),
    (0, 8, 16, (3, 1), (2, 1), (0, 0), (0, 0), (1, 1), 7, 20),
    (0, 8, 24, (3, 1), (1, 1), (0, 0), (0, 0), (1, 1), 3, 20),
    (1, 24, 16, (3, 1), (1, 1), (0, 0), (0, 0), (1, 1), 1, 20),
    (1, 16, 8, (4, 1), (2, 1), (1, 0), (1, 0), (1, 1), 3, 20),
    (1, 8, 8, (4, 1), (2, 1), (0, 0), (0, 0), (1, 1), 63, 20),
    (1, 8, 2, (3, 1), (1, 1), (1, 0), (0, 0), (1, 1), 33, 20),
    # odd channel counts (DSUNet 45 / 135), dilation, wide images
    (0, 45, 45, (1, 7), (1, 1), (0, 3), (0, 0), (1, 1), 5, 40),
    (1, 135, 90, (1, 7), (1, 1), (0, 3), (0, 0), (1, 1), 4, 34),
    (0, 4, 4, (3, 3), (1, 1), (2, 3), (0, 0), (2, 3), 17, 41),
    (1, 4, 4, (3, 3), (1, 1), (2, 2), (0, 0), (2, 2), 17, 41),
    (0, 3, 5, (3, 3), (1, 1), (1, 1), (0, 0), (1, 1), 1, 1),
    (1, 3, 5, (2, 2), (2, 2), (0, 0), (0, 0), (1, 1), 1, 1),
    (1, 4, 4, (1, 1), (2, 2), (0, 0), (1, 1), (1, 1), 6, 7),
]


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "{}c{}-{}k{}x{}s{}x{}_{}x{}".format(
    "t" if g[0] else "", g[1], g[2], g[3][0], g[3][1], g[4][0], g[4][1], g[8], g[9]))
def test_conv_bn_act_matches_torch_cpu(device, geom):
    import unet_amd
    from unet_amd.fused import FusedConv
    tr, cin, cout, k, s, p, op, d, H, W = geom
    if tr:
        conv = nn.ConvTranspose2d(cin, cout, k, s, p, output_padding=op, dilation=d)
    else:
        conv = nn.Conv2d(cin, cout, k, s, p, dilation=d)
    bn = nn.BatchNorm2d(cout)
    act = nn.LeakyReLU()
    seq = nn.Sequential(conv, bn, act).eval()
    _fill(seq, 1234 + cin * 7 + cout)
    x = _seeded((3, cin, H, W), 99 + H)
    with torch.no_grad():
        want = seq(x)
    dev = nn.Sequential(conv, bn, act).to(device)
    got = FusedConv(dev[0], dev[1], dev[2])(x.to(device))
    assert_close(got, want, what="conv+bn+leaky")
    seq.cpu()


# the MFMA rows of GEOMS with 33+ output channels (64-channel blocks), where both kernel families are built
WIDE_GEOMS = [g for g in GEOMS if g[2] > 32]


@pytest.mark.parametrize("row_tiles", [1, 2, 4, 5, 6])
@pytest.mark.parametrize("geom", WIDE_GEOMS, ids=lambda g: "{}c{}-{}k{}x{}s{}x{}_{}x{}".format(
    "t" if g[0] else "", g[1], g[2], g[3][0], g[3][1], g[4][0], g[4][1], g[8], g[9]))
def test_row_tile_kernel_families(device, geom, row_tiles, monkeypatch):
    """The library picks one row tile per wave for small grids, two or four otherwise, and five or six for
    large single-column-phase grids; all families on the same shapes (every test shape here is a
    small grid, so the default run alone would never reach the four-tile kernels), with a concatenated second
    source on the transposed ones."""
    import unet_amd.fused as fused
    from unet_amd.fused import FusedConv
    monkeypatch.setattr(fused, "ROW_TILES", row_tiles)
    tr, cin, cout, k, s, p, op, d, H, W = geom
    conv = (nn.ConvTranspose2d(cin, cout, k, s, p, output_padding=op, dilation=d) if tr
            else nn.Conv2d(cin, cout, k, s, p, dilation=d))
    seq = nn.Sequential(conv, nn.BatchNorm2d(cout), nn.LeakyReLU()).eval()
    _fill(seq, 4321 + cin * 5 + cout)
    x = _seeded((3, cin, H, W), 77 + W)
    with torch.no_grad():
        want = seq(x)
    seq.to(device)
    f = FusedConv(seq[0], seq[1], seq[2])
    if tr and cin >= 4:
        c0 = cin // 2 if (cin // 2) % 2 == 0 else cin // 2 + 1
        got = f(x[:, :c0].contiguous().to(device), x[:, c0:].contiguous().to(device))
    else:
        got = f(x.to(device))
    assert_close(got, want, what="row_tiles={}".format(row_tiles))
    seq.cpu()


def test_row_tile_request_is_validated(device):
    import unet_amd.fused as fused
    from unet_amd import _lib
    from unet_amd.fused import FusedConv
    conv = nn.Conv2d(4, 64, (5, 3), (2, 1), (2, 1)).to(device).eval()
    x = torch.zeros(1, 4, 9, 16, device=device)
    try:
        fused.ROW_TILES = 3
        with pytest.raises(_lib.UnetmiError):
            FusedConv(conv)(x)
    finally:
        fused.ROW_TILES = 0


ACTS = [nn.LeakyReLU(), nn.PReLU(), nn.PReLU(12), nn.ReLU(), nn.SiLU(), nn.Softplus(), nn.Sigmoid(),
        nn.Tanh(), nn.GELU(), nn.Identity()]


@pytest.mark.parametrize("act", ACTS, ids=lambda a: type(a).__name__ + (
    str(a.weight.numel()) if isinstance(a, nn.PReLU) else ""))
@pytest.mark.parametrize("act_first", [False, True])
def test_epilogue_variants(device, act, act_first):
    """Every activation, both epilogue orders, with gate and residual."""
    import copy
    from unet_amd.fused import FusedConv
    act = copy.deepcopy(act)
    conv = nn.Conv2d(6, 12, (3, 3), 1, (1, 1))
    bn = nn.BatchNorm2d(12)
    mods = nn.ModuleList([conv, bn, act]).eval()
    _fill(mods, 7)
    x = _seeded((2, 6, 9, 37), 5, 2.0)
    gate = torch.rand(2, 12, generator=torch.Generator().manual_seed(3))
    res = _seeded((2, 12, 9, 37), 6)
    with torch.no_grad():
        y = conv(x)
        y = bn(act(y)) if act_first else act(bn(y))
        want = y * gate[:, :, None, None] + res
    mods.to(device)
    got = FusedConv(mods[0], mods[1], mods[2], act_first=act_first)(
        x.to(device), gate=gate.to(device), residual=res.to(device))
    assert_close(got, want, what=type(act).__name__)
    mods.cpu()


def test_concat_and_add_inputs(device):
    """Decoder input = cat([p, skip]) (UNet.py:106) and y + res (ResUNet.py:209)
    read inside the kernel."""
    from unet_amd import _lib
    from unet_amd.fused import FusedConv
    tconv = nn.ConvTranspose2d(10 + 6, 8, (5, 3), (2, 2), (2, 1), output_padding=(0, 1))
    bn = nn.BatchNorm2d(8)
    mods = nn.ModuleList([tconv, bn]).eval()
    _fill(mods, 21)
    p, skip = _seeded((2, 10, 7, 33), 1), _seeded((2, 6, 7, 33), 2)
    with torch.no_grad():
        want = torch.nn.functional.leaky_relu(bn(tconv(torch.cat([p, skip], 1))), 0.01)
    mods.to(device)
    got = FusedConv(mods[0], mods[1], nn.LeakyReLU())(p.to(device), skip.to(device))
    assert_close(got, want, what="concat")
    mods.cpu()

    conv = nn.ConvTranspose2d(8, 8, (4, 1), (2, 1), (1, 0), output_padding=(1, 0)).eval()
    _fill(conv, 22)
    a, b = _seeded((2, 8, 15, 21), 3), _seeded((2, 8, 15, 21), 4)
    with torch.no_grad():
        want = conv(a + b)
    conv.to(device)
    got = FusedConv(conv)(a.to(device), b.to(device), in_mode=_lib.IN_ADD)
    assert_close(got, want, what="add")


def test_full_width_layers(device):
    """The two widest table rows at full channel count on a cut-down image:
    enc_2 (64->64, 7x5, s2x2) and dec_7 (128->64 transposed 7x5, s2x2 with concat)."""
    from unet_amd.fused import FusedConv
    enc = nn.Sequential(nn.Conv2d(64, 64, (7, 5), (2, 2), (3, 2)), nn.BatchNorm2d(64)).eval()
    _fill(enc, 31)
    x = _seeded((2, 64, 65, 141), 8)
    with torch.no_grad():
        want = torch.nn.functional.leaky_relu(enc(x), 0.01)
    enc.to(device)
    got = FusedConv(enc[0], enc[1], nn.LeakyReLU())(x.to(device))
    assert_close(got, want, what="enc_2")

    dec = nn.Sequential(nn.ConvTranspose2d(128, 64, (7, 5), (2, 2), (3, 2)), nn.BatchNorm2d(64)).eval()
    _fill(dec, 32)
    p, s = _seeded((2, 64, 33, 71), 9), _seeded((2, 64, 33, 71), 10)
    with torch.no_grad():
        want = torch.nn.functional.leaky_relu(dec(torch.cat([p, s], 1)), 0.01)
    dec.to(device)
    got = FusedConv(dec[0], dec[1], nn.LeakyReLU())(p.to(device), s.to(device))
    assert_close(got, want, what="dec_7")


def test_empty_batch_and_bad_input(device):
    from unet_amd.fused import FusedConv
    conv = nn.Conv2d(4, 8, 3, 1, 1).to(device).eval()
    out = FusedConv(conv)(torch.zeros(0, 4, 5, 5, device=device))
    assert tuple(out.shape) == (0, 8, 5, 5)
    with pytest.raises(RuntimeError):
        FusedConv(conv)(torch.zeros(1, 5, 5, 5, device=device))


@pytest.mark.parametrize("act", ["PReLU", "LeakyReLU"])
@pytest.mark.parametrize("shape", [(2, 64, 9, 70), (1, 64, 130, 33)])
def test_pointwise_64_encoder_block(device, act, shape):
    """Encoder(64, 64, (1,1)) -- ResUNet's res_paths -- takes the one-pass pointwise kernel; checked
    against the oracle block (conv -> BN -> act order), ragged tiles."""
    import oracle
    import unet_amd
    from unet_amd.seeding import normal
    ref = oracle.Encoder(64, 64, (1, 1), (1, 1), activation=act).eval()
    unet_amd.seeded_fill_(ref, seed=61)
    dev = unet_amd.Encoder(64, 64, (1, 1), (1, 1), activation=act)
    dev.load_state_dict(ref.state_dict(), strict=True)
    dev.eval().to(device)
    x = normal("pw64/x", shape, seed=61)
    with torch.no_grad():
        assert_close(dev(x.to(device)), ref(x), what="pointwise 64->64 block")


UPCONV_CASES = [
    # (N, H, W, pad, out_pad, activation, act_first, gate, residual)
    (3, 7, 29, 1, 1, "prelu", True, False, True),          # ResUNet's decoder layer (PReLU then BatchNorm, skip added)
    (8, 15, 130, 1, 1, "prelu", True, False, True),        # batch multiple of 8: the by-XCD tile order
    (2, 1, 64, 0, 0, "prelu64", False, False, False),      # one input row, per-channel slopes, norm then activation
    (1, 5, 1, 0, 1, "leaky", False, True, True),           # one column, gate and added tensor
    (2, 6, 67, 1, 0, "relu", False, False, True),
    (1, 9, 70, 0, 0, "none", False, False, False),
    (2, 4, 33, 1, 1, "silu", True, True, True),            # an activation without a slope form
    (16, 31, 501, 1, 1, "prelu", True, False, True),       # the benchmark's width, many tiles per workgroup
]


@pytest.mark.parametrize("case", UPCONV_CASES, ids=lambda c: "N{}H{}W{}p{}op{}{}".format(*c[:6]))
def test_frequency_doubling_decoder_layer(device, case):
    """ConvTranspose2d(64, 64, (4,1), stride (2,1)) -- ResUNet.py:88-89, the layer that carries the folded skip add:
    both paddings and output paddings, ragged and degenerate sizes, every epilogue form, against the torch CPU ops."""
    from unet_amd.fused import FusedConv
    N, H, W, pad, opad, act_name, act_first, with_gate, with_res = case
    act = {"prelu": nn.PReLU(), "prelu64": nn.PReLU(64), "leaky": nn.LeakyReLU(0.2), "relu": nn.ReLU(),
           "none": nn.Identity(), "silu": nn.SiLU()}[act_name]
    conv = nn.ConvTranspose2d(64, 64, (4, 1), (2, 1), (pad, 0), output_padding=(opad, 0))
    bn = nn.BatchNorm2d(64)
    mods = nn.ModuleList([conv, bn, act]).eval()
    _fill(mods, 400 + H)
    if act_name == "prelu64":
        with torch.no_grad():
            act.weight.copy_(torch.linspace(-0.3, 0.6, 64))
    x = _seeded((N, 64, H, W), 31 + W, 1.5)
    with torch.no_grad():
        y = conv(x)
        y = bn(act(y)) if act_first else act(bn(y))
    gate = torch.rand(N, 64, generator=torch.Generator().manual_seed(4)) if with_gate else None
    res = _seeded(tuple(y.shape), 8) if with_res else None
    want = y
    if gate is not None:
        want = want * gate[:, :, None, None]
    if res is not None:
        want = want + res
    mods.to(device)
    got = FusedConv(mods[0], mods[1], mods[2], act_first=act_first)(
        x.to(device), gate=None if gate is None else gate.to(device), residual=None if res is None else res.to(device))
    assert tuple(got.shape) == tuple(want.shape)
    assert_close(got, want, what="frequency-doubling layer " + act_name)
    mods.cpu()
