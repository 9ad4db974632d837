"""DSUNet bf16-operand forward: time and error for several choices of which layers take two weight halves."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd
from unet_amd.blocked import BlockedConv
from unet_amd.seeding import normal, uniform

ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8)).eval()
unet_amd.seeded_fill_(ref, seed=41)
dev = unet_amd.DSUNet(unet_amd.dsunet_architecture(45, 8))
dev.load_state_dict(ref.state_dict(), strict=True)
dev.eval().cuda()
sf = normal("dsunet/sfbf", (2, 8, 257, 32), seed=42).to(torch.bfloat16)
SV = normal("dsunet/SV", (2, 4, 514), seed=42)
ang = uniform("dsunet/theta", (2,), 0.0, 6.283185, seed=42)
theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
big = torch.randn(32, 8, 257, 512).to(torch.bfloat16).cuda()
SVb, thb = torch.randn(32, 4, 514).cuda(), torch.randn(32, 2).cuda()
with torch.no_grad():
    want = torch.view_as_real(ref(sf.float(), SV, theta))
    for label, enc2, dec2, head2 in (("none", (), (), 1), ("head", (), (), 2), ("enc0+head", (0,), (), 2),
                                     ("enc0+dec9+head", (0,), (9,), 2), ("enc0,1+dec8,9+head", (0, 1), (8, 9), 2)):
        built = dict(enc=[BlockedConv(m.conv, m.bn, m.relu, halves=2 if i in enc2 else 1) for i, m in enumerate(dev.encoders)],
                     dec=[BlockedConv(m.transconv, m.bn, m.relu, halves=2 if i in dec2 else 1) for i, m in enumerate(dev.decoders)],
                     head=BlockedConv(dev.linear, None, None, halves=head2, nchw_out=True))
        object.__setattr__(dev, "_blocked", built)
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
