"""Error of the 16-bit-operand paths (a bfloat16 input, fp16 or bfloat16 operands inside) at the benchmark's full
length against the fp32 CPU oracle: UNet (T=509), UNet20 (T=509), and their time per batch."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd
from unet_amd.seeding import normal, uniform

dev = torch.device("cuda:0")


def errs(got, want):
    if got.is_complex():
        got, want = torch.view_as_real(got), torch.view_as_real(want)
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    return ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32)).eval()
    unet_amd.seeded_fill_(ref, seed=5)
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    net.load_state_dict(ref.state_dict(), strict=True)
    net.eval().to(dev)
    x = normal("unet_bf16_full/x", (2, 2, 257, 509), seed=6).to(torch.bfloat16)
    want = ref(x.float())
    big = torch.randn(64, 2, 257, 509).to(torch.bfloat16).to(dev)
    print("UNet  fp32 input                    : {:.2f} ms / batch 64".format(timed(lambda: net(big.float()))))
    for label, matrix, fmt in (("bf16 storage (NCHW, fp32 products)", False, "fp16"), ("fp16 operands behind enc0", True, "fp16"),
                               ("bf16 operands", True, "bf16")):
        net.bf16_matrix, net.operand_format = matrix, fmt
        l2, mx = errs(net(x.to(dev)), want)
        print("UNet  {:34s}: L2 {:.2e} max-norm {:.2e}  {:.2f} ms / batch 64".format(label, l2, mx, timed(lambda: net(big))))
    del net, big
    ref20 = oracle.UNet20().eval()
    unet_amd.seeded_fill_(ref20, seed=21)
    net20 = unet_amd.UNet20()
    net20.load_state_dict(ref20.state_dict(), strict=True)
    ref20.T = net20.T = 509
    net20.eval().to(dev)
    x20 = uniform("unet20_bf16_full/x", (1, 14, 257, 509), 0.0, 1.0, seed=22).to(torch.bfloat16)
    want20 = ref20(x20.float())
    big20 = torch.rand(32, 14, 257, 509).to(torch.bfloat16).to(dev)
    print("UNet20 fp32 input                   : {:.2f} ms / batch 32".format(timed(lambda: net20(big20.float()))))
    for label, matrix, fmt in (("bf16 storage (NCHW, fp32 products)", False, "fp16"), ("fp16 operands behind enc0", True, "fp16"),
                               ("bf16 operands", True, "bf16")):
        net20.bf16_matrix, net20.operand_format = matrix, fmt
        l2, mx = errs(net20(x20.to(dev)), want20)
        print("UNet20 {:33s}: L2 {:.2e} max-norm {:.2e}  {:.2f} ms / batch 32".format(label, l2, mx, timed(lambda: net20(big20))))
