"""Do 128-byte-aligned rows speed up the big fp32 convolution layers?  The headline layers at their own widths
(509 / 255 columns: row pitch not a multiple of 128 bytes) and at the next aligned width (512 / 256), batch 64."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, ".")
from unet_amd.fused import FusedConv
dev = torch.device("cuda", 0)


def run(name, conv, H, W, N=64, c1=0):
    bn = nn.BatchNorm2d(conv.out_channels).eval(); act = nn.LeakyReLU(0.01)
    m = nn.Sequential(conv, bn, act).to(dev)
    f = FusedConv(m[0], m[1], m[2])
    cin = conv.in_channels
    x = torch.randn(N, cin - c1, H, W, device=dev)
    x2 = torch.randn(N, c1, H, W, device=dev) if c1 else None
    for _ in range(3): y = f(x, x2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = f(x, x2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    k = conv.kernel_size; tr = isinstance(conv, nn.ConvTranspose2d)
    macs = cin * conv.out_channels * k[0] * k[1] * (H * W if tr else y.shape[2] * y.shape[3]) * N
    print("%-34s W=%d  %8.1f us  %6.1f TFLOP/s" % (name, W, dt * 1e6, 2 * macs / dt / 1e12), flush=True)


for W in (509, 512):
    run("enc_1 conv(1,7) 32->64", nn.Conv2d(32, 64, (1, 7), 1, (0, 3)), 257, W)
    run("enc_2 conv(7,5)s(2,2) 64->64", nn.Conv2d(64, 64, (7, 5), (2, 2), (3, 2)), 257, W)
    run("dec_8 tconv(1,7) 64+64->64", nn.ConvTranspose2d(128, 64, (1, 7), 1, (0, 3)), 257, W, c1=64)
for W in (255, 256):
    run("enc_3 conv(7,5)s(2,1) 64->64", nn.Conv2d(64, 64, (7, 5), (2, 1), (3, 2)), 129, W)
    run("dec_6 tconv(7,5)s(2,1) 128->64", nn.ConvTranspose2d(128, 64, (7, 5), (2, 1), (3, 2)), 65, W, c1=64)
    run("dec_7 tconv(7,5)s(2,2) 128->64", nn.ConvTranspose2d(128, 64, (7, 5), (2, 2), (3, 2)), 129, W, c1=64)
