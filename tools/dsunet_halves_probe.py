"""DSUNet bf16-operand forward: time and error for several choices of which layers take two weight halves."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd
from unet_amd.seeding import normal, uniform

ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8)).eval()
unet_amd.seeded_fill_(ref, seed=41)
dev = unet_amd.DSUNet(unet_amd.dsunet_architecture(45, 8))
dev.load_state_dict(ref.state_dict(), strict=True)
dev.eval().cuda()
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"        # the full-length test's inputs (T=512) instead of T=32
if FULL:
    ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8)).eval()
    unet_amd.seeded_fill_(ref, seed=44)
    dev.load_state_dict(ref.state_dict(), strict=True)
    sf = torch.cat([normal("dsunet/sf512", (1, 8, 257, 512), seed=44),
                    normal("dsunet/sf512b", (1, 8, 257, 512), seed=45)]).to(torch.bfloat16)
    SV = torch.cat([normal("dsunet/SV512", (1, 4, 514), seed=44), normal("dsunet/SV512b", (1, 4, 514), seed=45)])
    ang = torch.cat([uniform("dsunet/theta512", (1,), 0.0, 6.283185, seed=44),
                     uniform("dsunet/theta512b", (1,), 0.0, 6.283185, seed=45)])
else:
    sf = normal("dsunet/sfbf", (2, 8, 257, 32), seed=42).to(torch.bfloat16)
    SV = normal("dsunet/SV", (2, 4, 514), seed=42)
    ang = uniform("dsunet/theta", (2,), 0.0, 6.283185, seed=42)
theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
big = torch.randn(32, 8, 257, 512).to(torch.bfloat16).cuda()
SVb, thb = torch.randn(32, 4, 514).cuda(), torch.randn(32, 2).cuda()
with torch.no_grad():
    want = torch.view_as_real(ref(sf.float(), SV, theta))
    for fmt_label, enc2 in (("fp16 behind enc0, enc0 one half", ()), ("fp16 behind enc0, enc0 two halves", (0,))):
        dev.bf16_two_half_encoders, dev.operand_format = enc2, "fp16"
        got = torch.view_as_real(dev(sf.cuda(), SV.cuda(), theta.cuda()).cpu())
        l2, mx = ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()
        for _ in range(3):
            dev(big, SVb, thb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dev(big, SVb, thb)
        torch.cuda.synchronize()
        print("{:34s} L2 {:.2e} max-norm {:.2e}  {:.2f} ms / batch 32".format(fmt_label, l2, mx, (time.perf_counter() - t0) * 100))
    for label, enc2, dec2 in (("none", (), ()), ("enc0", (0,), ()), ("dec9", (), (9,)), ("enc0+dec9", (0,), (9,)),
                              ("enc0,1+dec9", (0, 1), (9,)), ("enc0+dec8,9", (0,), (8, 9)),
                              ("enc0,1+dec8,9", (0, 1), (8, 9)), ("all", tuple(range(10)), tuple(range(10)))):
        dev.bf16_two_half_encoders, dev.bf16_two_half_decoders = enc2, dec2
        dev.operand_format = "bf16"
        object.__setattr__(dev, "_blocked", None)
        got = torch.view_as_real(dev(sf.cuda(), SV.cuda(), theta.cuda()).cpu())
        l2, mx = ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()
        for _ in range(3):
            dev(big, SVb, thb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dev(big, SVb, thb)
        torch.cuda.synchronize()
        print("{:22s} L2 {:.2e} max-norm {:.2e}  {:.2f} ms / batch 32".format(label, l2, mx, (time.perf_counter() - t0) * 100))
