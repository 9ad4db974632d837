import sys, time, torch, torch.nn as nn
sys.path.insert(0, "/root/repo")
from unet_amd.fused import FusedConv
dev = torch.device("cuda", 0)
def run(name, conv, H, W, N=32, add=False):
    cin = conv.in_channels
    bn = nn.BatchNorm2d(conv.out_channels).eval(); act = nn.PReLU()
    m = nn.Sequential(conv, bn, act).to(dev)
    f = FusedConv(m[0], m[1], m[2])
    x = torch.randn(N, cin, H, W, device=dev)
    for _ in range(3): y = f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = f(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    k = conv.kernel_size
    tr = isinstance(conv, nn.ConvTranspose2d)
    macs = cin * conv.out_channels * k[0] * k[1] * (x.shape[2] * x.shape[3] if tr else y.shape[2] * y.shape[3]) * N
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print("%-28s in %s out %s  %8.1f us  %6.1f TFLOP/s  %5.2f TB/s" % (name, tuple(x.shape[1:]), tuple(y.shape[1:]), dt * 1e6, 2 * macs / dt / 1e12, gb / dt / 1e3), flush=True)
for H in (128, 63, 31):
    run("tconv(4,1)s(2,1) H=%d" % H, nn.ConvTranspose2d(64, 64, (4, 1), (2, 1), (1, 0), output_padding=(1, 0)), H, 501)
for H in (257, 128, 63):
    run("conv(3,3)s(2,1) H=%d" % H, nn.Conv2d(64, 64, (3, 3), (2, 1), (0, 1)), H, 501)
run("conv(3,1)s(2,1) 64->128", nn.Conv2d(64, 128, (3, 1), (2, 1), (0, 0)), 7, 501)
run("conv(3,1) 128->384", nn.Conv2d(128, 384, (3, 1), (1, 1), (0, 0)), 3, 501)
run("tconv(3,1) 384->128", nn.ConvTranspose2d(384, 128, (3, 1), (1, 1), (0, 0)), 1, 501)
print("--- scaling of the small layers")
for N in (1, 8, 32):
    run("conv(3,1) 128->384 N=%d" % N, nn.Conv2d(128, 384, (3, 1), (1, 1), (0, 0)), 3, 501, N=N)
for H in (3, 10, 34):
    run("conv(3,1) 128->384 H=%d" % H, nn.Conv2d(128, 384, (3, 1), (1, 1), (0, 0)), H, 501)
run("conv(3,1) 128->64 H=3", nn.Conv2d(128, 64, (3, 1), (1, 1), (0, 0)), 3, 501)
run("conv(3,1) 128->128 H=3", nn.Conv2d(128, 128, (3, 1), (1, 1), (0, 0)), 3, 501)
run("conv(3,1) 128->384 W=125", nn.Conv2d(128, 384, (3, 1), (1, 1), (0, 0)), 3, 125)
print("--- one-column kernels of the other networks")
run("UNet20 dec_9 tconv(7,1) 96->64", nn.ConvTranspose2d(96, 64, (7, 1), (1, 1), (3, 0)), 257, 509)
run("conv(1,1) 64->32 head", nn.Conv2d(64, 32, 1), 257, 509)
