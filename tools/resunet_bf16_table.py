"""ResUNet forward at batch 32: time of the fp32 path, of a bfloat16 input on the default path (widened at the
boundary) and of the opt-in all-bfloat16 storage, at the benchmark width T=501 (odd: bfloat16 rows are not 4-byte
aligned, so the 16-byte LDS-DMA kernels cannot serve them) and at T=504 (what a row pitch padded to 8 elements would
give every kernel)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd

dev = torch.device("cuda:0")
net = unet_amd.ResUNet()
unet_amd.seeded_fill_(net, seed=31)
net.eval().to(dev)


def timed(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


with torch.no_grad():
    for T in (501, 504):
        x = torch.randn(32, 4, 257, T, device=dev)
        xb = x.to(torch.bfloat16)
        net.bf16_storage = False
        a = timed(lambda: net(x))
        b = timed(lambda: net(xb))
        net.bf16_storage = True
        c = timed(lambda: net(xb))
        print("T={}: fp32 {:.2f} ms, bf16 in/out (fp32 inside) {:.2f} ms, bf16 storage {:.2f} ms".format(T, a, b, c), flush=True)
