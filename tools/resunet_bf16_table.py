"""ResUNet forward at batch 32: time of the fp32 path and of a bfloat16 input on its paths -- fp32 products (bfloat16
at the boundary only), bfloat16 operands in the strided convolutions with one / two weight halves, all-bfloat16
storage -- at the benchmark width T=501 and at T=504 (what a row pitch padded to 8 elements would give the storage
path), plus the error of each against the fp32 CPU oracle at T=501, batch 1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd

dev = torch.device("cuda:0")
ref = oracle.ResUNet().eval()
unet_amd.seeded_fill_(ref, seed=31)
net = unet_amd.ResUNet()
net.load_state_dict(ref.state_dict(), strict=True)
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


def configure(operands, storage):
    net.bf16_operands, net.bf16_storage = operands, storage


MODES = (("fp32 products", 0, False), ("bf16 operands, 1 half", 1, False), ("bf16 operands, 2 halves", 2, False),
         ("bf16 storage", 0, True))
with torch.no_grad():
    xe = torch.randn(1, 4, 257, 501).to(torch.bfloat16)
    want = ref(xe.float())
    for label, op, st in MODES:
        configure(op, st)
        got = net(xe.to(dev)).float().cpu()
        print("{:26s} L2 {:.2e}  max-norm {:.2e}".format(label, ((got - want).norm() / want.norm()).item(),
                                                       ((got - want).abs().max() / want.abs().max()).item()), flush=True)
    for T in (501, 504):
        x = torch.randn(32, 4, 257, T, device=dev)
        xb = x.to(torch.bfloat16)
        configure(0, False)
        line = "T={}: fp32 {:.2f} ms".format(T, timed(lambda: net(x)))
        for label, op, st in MODES:
            configure(op, st)
            line += ", bf16 input / {} {:.2f} ms".format(label, timed(lambda: net(xb)))
        print(line, flush=True)
