"""ResUNet's strided convolutions at batch 32: the fp32 MFMA path (FusedConv on NCHW fp32) against
fp32 NCHW -> blocked bf16 (unetmi_blocked_from_nchw) -> bf16-operand conv writing fp32 NCHW (conv2d_b16, nchw_out)."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, ".")
from unet_amd.fused import FusedConv
from unet_amd.blocked import BlockedConv, BlockedTensor
dev = torch.device("cuda", 0)


def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e6


def run(name, conv, H, W=501, N=32):
    bn = nn.BatchNorm2d(conv.out_channels).eval(); act = nn.PReLU()
    m = nn.Sequential(conv, bn, act).to(dev)
    x = torch.randn(N, conv.in_channels, H, W, device=dev)
    f = FusedConv(m[0], m[1], m[2])
    b = BlockedConv(m[0], m[1], m[2], halves=1, nchw_out=True)
    b2 = BlockedConv(m[0], m[1], m[2], halves=2, nchw_out=True)
    t_f = timed(lambda: f(x))
    t_c = timed(lambda: BlockedTensor.from_nchw(x))
    xb = BlockedTensor.from_nchw(x)
    t_b = timed(lambda: b(xb))
    t_b2 = timed(lambda: b2(xb))
    y0, y1, y2 = f(x), b(xb), b2(xb)
    e1 = ((y1 - y0).abs().max() / y0.abs().max()).item(); e2 = ((y2 - y0).abs().max() / y0.abs().max()).item()
    print("%-30s H=%3d fp32 %7.1f us | to_blocked %6.1f + conv_b16 %6.1f (x2: %6.1f) = %7.1f us  err %.1e / %.1e" % (
        name, H, t_f, t_c, t_b, t_b2, t_c + t_b, e1, e2), flush=True)


for H in (257, 128, 63, 31, 15):
    run("conv(3,3)s(2,1) 64->64", nn.Conv2d(64, 64, (3, 3), (2, 1), (0, 1)), H)
for H in (128, 63, 31, 15, 7):
    run("tconv(4,1)s(2,1) 64->64", nn.ConvTranspose2d(64, 64, (4, 1), (2, 1), (1, 0), output_padding=(1, 0)), H)
run("conv(3,1)s(2,1) 64->128", nn.Conv2d(64, 128, (3, 1), (2, 1), (0, 0)), 7)
run("tconv(4,1)s(2,1) 128->64", nn.ConvTranspose2d(128, 64, (4, 1), (2, 1), (1, 0), output_padding=(1, 0)), 3)
run("1x1 64->64", nn.Conv2d(64, 64, 1), 257)
