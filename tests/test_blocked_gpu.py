"""bfloat16-operand convolution on the channel-blocked layout (csrc/conv_b16.hip) against the CPU
restatement: the same layer in fp32 torch ops on the bfloat16-rounded input.

Tolerances: with two weight halves (hi + rounding remainder) the products equal fp32-weight products up
to 2^-17 relative, so the result is held to the fp32 bar of 1e-4 before the output rounding; the output
itself is bfloat16, so the comparison is made against the reference rounded the same way, with one
bfloat16 ulp (2^-8 relative, on the tensor's scale) of slack for values that straddle a rounding
boundary.  With one half (weights rounded to bfloat16) the reference uses the rounded weights too.
"""

import pytest
import torch
import torch.nn as nn

from util import rel_err
from unet_amd._lib import UnetmiError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda", 0)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rnd(t, fmt):
    """t rounded to the blocked tensor's 16-bit format"""
    return t.to(torch.float16 if fmt == "fp16" else torch.bfloat16).to(torch.float32)


def test_layout_round_trip(device):
    from unet_amd.blocked import BlockedTensor
    g = torch.Generator().manual_seed(1)
    for C in (1, 8, 16, 45, 90):
        x = torch.randn(2, C, 5, 37, generator=g)
        for src in (x, x.to(torch.bfloat16)):
            b = BlockedTensor.from_nchw(src.to(device))
            assert tuple(b.data.shape) == (2, 2 * ((C + 15) // 16), 5, 37, 8)
            assert torch.equal(b.to_nchw(torch.float32).cpu(), bf(x))
            assert torch.equal(b.to_nchw(torch.bfloat16).cpu(), x.to(torch.bfloat16))
            # padded channels are zero
            flat = b.data.permute(0, 1, 4, 2, 3).reshape(2, -1, 5, 37)
            assert flat[:, C:].abs().max().item() == 0 if flat.shape[1] > C else True
            # the fp16 form of the layout: same records, three more mantissa bits
            h = BlockedTensor.from_nchw(src.to(device), fmt="fp16")
            assert h.data.dtype == torch.float16 and h.fmt == "fp16" and b.fmt == "bf16"
            assert torch.equal(h.to_nchw(torch.float32).cpu(), src.float().to(torch.float16).float())
    # beyond the format's range: infinity (and NaN stays NaN), never a silently clipped value
    big = torch.tensor([[[[1e6, -1e6, 65504.0, 65519.0, float("nan")]]]])
    far = BlockedTensor.from_nchw(big.to(device), fmt="fp16").to_nchw(torch.float32).cpu()[0, 0, 0]
    assert far[:4].tolist() == [float("inf"), float("-inf"), 65504.0, 65504.0] and torch.isnan(far[4])


CASES = [
    # (C0, C1, K, kernel, stride, padding, output_padding, transposed, H, W)
    (8, 0, 45, (7, 1), (1, 1), (3, 0), (0, 0), False, 19, 70),
    (45, 0, 45, (1, 7), (1, 1), (0, 3), (0, 0), False, 11, 77),
    (45, 0, 90, (7, 5), (2, 2), (3, 2), (0, 0), False, 33, 70),
    (90, 0, 90, (7, 5), (2, 1), (3, 2), (0, 0), False, 33, 41),
    (90, 0, 90, (5, 3), (2, 2), (2, 1), (0, 0), False, 17, 34),
    (90, 0, 128, (5, 3), (2, 1), (2, 1), (0, 0), False, 5, 34),
    (128, 0, 90, (5, 3), (2, 1), (2, 1), (0, 0), True, 2, 32),
    (90, 90, 90, (5, 3), (2, 2), (2, 1), (0, 1), True, 3, 16),
    (90, 90, 90, (7, 5), (2, 1), (3, 2), (0, 0), True, 17, 40),
    (90, 45, 90, (7, 5), (2, 2), (3, 2), (0, 1), True, 9, 35),
    (90, 45, 90, (1, 7), (1, 1), (0, 3), (0, 0), True, 9, 70),
    (90, 8, 90, (7, 1), (1, 1), (3, 0), (0, 0), True, 21, 33),
    (16, 0, 32, (3, 3), (1, 1), (1, 2), (0, 0), False, 9, 40),
    (20, 12, 64, (3, 3), (1, 1), (1, 1), (0, 0), False, 12, 65),
    # more than 128 outputs: 128-channel chunks as a grid dimension (ResUNet's 384-channel layers, the LSTM's products)
    (128, 0, 384, (3, 1), (1, 1), (0, 0), (0, 0), False, 5, 70),
    (384, 0, 128, (3, 1), (1, 1), (0, 0), (0, 0), True, 3, 40),
    (48, 0, 256, (1, 1), (1, 1), (0, 0), (0, 0), False, 9, 33),
    # time stride 2 on odd and tiny widths, one and two column tiles, a time-only stride, an even kernel width
    (64, 0, 64, (7, 5), (2, 2), (3, 2), (0, 0), False, 21, 75),
    (32, 0, 40, (3, 3), (1, 2), (1, 1), (0, 0), False, 9, 141),
    (16, 0, 16, (1, 4), (1, 2), (0, 1), (0, 0), False, 3, 9),
    (16, 0, 16, (3, 5), (2, 2), (1, 2), (0, 0), False, 5, 2),
    # edges: one pixel, one channel, tiles hanging over every border, time-only stride, dilation via a wide pad
    (1, 0, 1, (1, 1), (1, 1), (0, 0), (0, 0), False, 1, 1),
    (3, 0, 2, (3, 3), (1, 1), (1, 1), (0, 0), False, 2, 3),
    (16, 0, 16, (1, 3), (1, 2), (0, 1), (0, 1), True, 4, 9),
    (17, 1, 33, (3, 1), (2, 1), (1, 0), (1, 0), True, 5, 7),
    (8, 8, 16, (5, 3), (2, 2), (2, 1), (1, 1), True, 1, 1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "C{}+{}_K{}_k{}x{}_s{}{}_{}".format(
    c[0], c[1], c[2], c[3][0], c[3][1], c[4][0], c[4][1], "T" if c[7] else "C"))
@pytest.mark.parametrize("halves", [1, 2])
def test_conv_b16_matches_cpu(device, case, halves):
    run_case(device, case, halves, 2)


@pytest.mark.parametrize("case", CASES, ids=lambda c: "C{}+{}_K{}_k{}x{}_s{}{}_{}".format(
    c[0], c[1], c[2], c[3][0], c[3][1], c[4][0], c[4][1], "T" if c[7] else "C"))
def test_conv_fp16_operands_match_cpu(device, case):
    """The same layers with fp16 operands (inputs, weight image) and an fp16 blocked output: same kernels on
    v_mfma_f32_32x32x16_f16, held to the fp16 rounding of the same reference."""
    run_case(device, case, 1, 2, fmt="fp16")


# Shapes large enough for the plan to take the two-rows-per-wave kernels (16-row workgroup tiles; every phase at
# least 16 rows high and 512 workgroups or more): a (7,1) layer over 135 channels, whose regular plan is a
# one-column-tile kernel, and a time-stride-2 transposed 7x5 layer with three channel tiles.
RT2_CASES = [
    (16, (90, 45, 90, (7, 1), (1, 1), (3, 0), (0, 0), True, 64, 512), "conv_b16_kernel_rt2<3, false>"),
    (28, (90, 90, 90, (7, 5), (2, 2), (3, 2), (0, 1), True, 33, 128), "conv_b16_kernel_pair_rt2<3, false>"),
]


@pytest.mark.parametrize("N,case,kernel", RT2_CASES, ids=["7x1_135ch", "7x5_s22_T"])
def test_conv_b16_two_rows_per_wave(device, N, case, kernel):
    from unet_amd import _lib
    from unet_amd.ops import b16_desc, conv_b16_kernel_name
    C0, C1, K, k, s, p, op, transposed, H, W = case
    geom = [K, k[0], k[1], s[0], s[1], p[0], p[1], 1, 1, op[0], op[1]]
    d = b16_desc(N, C0, C1, H, W, geom, int(transposed), _lib.ACT_LEAKY, 0, 0, 1, 1, False)
    assert conv_b16_kernel_name(d) == kernel, "the plan no longer takes the two-row kernel for this shape"
    run_case(device, case, 1, N, with_head=False)
    d16 = b16_desc(N, C0, C1, H, W, geom, int(transposed), _lib.ACT_LEAKY, 0, 0, 1, 1, False, True, True)
    assert conv_b16_kernel_name(d16) == kernel.replace("false>", "true>")
    run_case(device, case, 1, N, with_head=False, fmt="fp16")


def run_case(device, case, halves, N, with_head=True, fmt="bf16"):
    from unet_amd.blocked import BlockedConv, BlockedTensor
    C0, C1, K, k, s, p, op, transposed, H, W = case
    g = torch.Generator().manual_seed(hash(case) % 1000 + halves)
    C = C0 + C1
    if transposed:
        conv = nn.ConvTranspose2d(C, K, k, stride=s, padding=p, output_padding=op)
    else:
        conv = nn.Conv2d(C, K, k, stride=s, padding=p)
    bn = nn.BatchNorm2d(K).eval()
    act = nn.LeakyReLU(0.01)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (C * k[0] * k[1]) ** 0.5)
        conv.bias.copy_(torch.randn(K, generator=g) * 0.1)
        bn.weight.copy_(torch.rand(K, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(K, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(K, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(K, generator=g) + 0.5)
    x0 = torch.randn(N, C0, H, W, generator=g)
    x1 = torch.randn(N, C1, H, W, generator=g) if C1 else None
    gate = torch.rand(N, K, generator=g) + 0.25
    ulp = 2.0 ** -11 if fmt == "fp16" else 2.0 ** -8           # one rounding step of the stored output
    with torch.no_grad():
        xin = rnd(x0, fmt) if x1 is None else torch.cat([rnd(x0, fmt), rnd(x1, fmt)], 1)
        ref_conv = conv
        if halves == 1:
            ref_conv = type(conv)(C, K, k, stride=s, padding=p, **(dict(output_padding=op) if transposed else {}))
            ref_conv.weight.copy_(rnd(conv.weight, fmt))
            ref_conv.bias.copy_(conv.bias)
        want = act(bn(ref_conv(xin))) * gate[:, :, None, None]
    layer = BlockedConv(conv, bn, act, halves=halves, out_fmt=fmt)
    b0 = BlockedTensor.from_nchw(x0.to(device), fmt=fmt)
    b1 = BlockedTensor.from_nchw(x1.to(device), fmt=fmt) if x1 is not None else None
    try:
        got = layer(b0, b1, gate.to(device))
    except UnetmiError as e:
        # the doubled weight image of the two-half mode does not fit beside a stride-2 input image
        assert halves == 2 and "LDS images exceed" in str(e), str(e)
        pytest.skip("two weight halves: LDS images of this layer exceed 160 KB")
    assert got.shape == tuple(want.shape)
    got32 = got.to_nchw(torch.float32).cpu()
    scale = want.abs().max().item()
    assert got.fmt == fmt
    err = (got32 - rnd(want, fmt)).abs().max().item() / scale
    assert err <= ulp + 1e-4, "blocked conv vs rounded reference: {:.3e}".format(err)
    # the unrounded accumulator path: fp32 NCHW output holds the fp32 bar
    head = BlockedConv(conv, bn, act, halves=halves, nchw_out=True)
    got_f32 = head(b0, b1, gate.to(device)).cpu()
    e = rel_err(got_f32, want)
    assert e <= 1e-4, "fp32-output blocked conv: rel-err {:.3e}".format(e)
    # ... plus an added fp32 tensor, and with the input formed as a sum during the layout conversion
    res = torch.randn(want.shape, generator=g)
    got_res = head(b0, b1, gate.to(device), residual=res.to(device)).cpu()
    e = rel_err(got_res, want + res)
    assert e <= 1e-4, "fp32-output blocked conv + residual: rel-err {:.3e}".format(e)
    xa, xb_ = x0 * 0.25, x0 - x0 * 0.25
    summed = BlockedTensor.from_nchw(xa.to(device), addend=xb_.to(device), fmt=fmt)
    assert torch.equal(summed.data.cpu(), BlockedTensor.from_nchw((xa + xb_).to(device), fmt=fmt).data.cpu())
    # padded output channels are exact zeros (the next layer multiplies them by zero weights)
    raw = got.data.permute(0, 1, 4, 2, 3).reshape(N, -1, *got.data.shape[2:4])
    if raw.shape[1] > K:
        assert raw[:, K:].float().abs().max().item() == 0.0
    if not with_head:
        return
    if K > 128:                                       # a layer of several 128-channel chunks takes no fused head
        from unet_amd import _lib
        with pytest.raises(UnetmiError, match="multiples of 128"):
            layer.with_head(b0, b1, nn.Conv2d(K, 2, 1).to(device), _lib.ACT_TANH, gate=gate.to(device))
        return
    # the layer + a 1x1 head + squash as one launch (the layer's result never leaves the fp32 accumulators):
    # planar and pixel-interleaved output, 1..4 head channels
    from unet_amd import _lib
    P = 1 + (hash(case) % 4)
    head_conv = nn.Conv2d(K, P, 1, bias=(hash(case) % 3) != 0)
    with torch.no_grad():
        head_conv.weight.copy_(torch.randn(head_conv.weight.shape, generator=g) / K ** 0.5)
        if head_conv.bias is not None:
            head_conv.bias.copy_(torch.randn(P, generator=g) * 0.1)
        want_head = torch.tanh(head_conv(want))
    if transposed and s[1] == 2:                      # the fused head is not built for time-stride-2 transposed layers
        with pytest.raises(UnetmiError, match="time-stride-2"):
            layer.with_head(b0, b1, head_conv.to(device), _lib.ACT_TANH, gate=gate.to(device))
        return
    for interleaved in (False, True):
        m = layer.with_head(b0, b1, head_conv.to(device), _lib.ACT_TANH, interleaved=interleaved, gate=gate.to(device)).cpu()
        head_conv.cpu()
        if interleaved:
            assert tuple(m.shape) == (N,) + tuple(want.shape[2:]) + (P,)
            m = m.permute(0, 3, 1, 2)
        e = rel_err(m, want_head)
        assert e <= 1e-4, "layer + head ({} channels, interleaved={}): rel-err {:.3e}".format(P, interleaved, e)


def test_conv_b16_empty_batch(device):
    from unet_amd.blocked import BlockedConv, BlockedTensor
    conv = nn.Conv2d(16, 16, 3, padding=1)
    x = BlockedTensor(torch.zeros(0, 2, 5, 7, 8, dtype=torch.bfloat16, device=device), 16)
    y = BlockedConv(conv.to(device), None, None)(x)
    assert y.shape == (0, 16, 5, 7)


def test_conv_b16_rejects(device):
    import ctypes
    from unet_amd import _lib
    from unet_amd.blocked import _desc
    d = _desc(1, 16, 0, 8, 8, [200, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0], 0, 0, 0, 0, 0, 1, False)
    assert _lib.lib.unetmi_conv2d_b16_packed_weight_bytes(ctypes.byref(d)) == 0
    assert "multiples of 128" in _lib.last_error()
