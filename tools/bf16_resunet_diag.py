"""Where ResUNet's bfloat16-storage error comes from: the same bf16 input with (a) every tensor bf16, (b) the
ResBlocks computed and stored in fp32 (bf16 at their boundaries), (c) only the ResBlocks in bf16."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd
from unet_amd.UNet_m import ResBlock
from unet_amd.seeding import normal

ref = oracle.ResUNet().eval()
unet_amd.seeded_fill_(ref, seed=31)
dev = unet_amd.ResUNet()
dev.load_state_dict(ref.state_dict(), strict=True)
dev.eval().cuda()
orig = ResBlock.forward


def norms(got, want):
    got, want = got.float().cpu(), want.float()
    return ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()


for T in (24, 25, 64):
    x = normal("resunet/xbf", (2, 4, 257, T), seed=35).to(torch.bfloat16)
    with torch.no_grad():
        want = ref(x.float())
        ResBlock.forward = orig
        a = dev(x.cuda())
        ResBlock.forward = lambda self, t, *args, **kw: orig(self, t.float(), *args, **kw).to(torch.bfloat16)
        b = dev(x.cuda())
        ResBlock.forward = lambda self, t, *args, **kw: orig(self, t.to(torch.bfloat16), *args, **kw).float()
        c = dev(x.cuda().float())
        ResBlock.forward = orig
        d = dev(x.cuda().float())
    print("T={}: all bf16 L2 {:.2e} max {:.2e} | ResBlocks fp32 L2 {:.2e} max {:.2e} | only ResBlocks bf16 L2 {:.2e} max {:.2e}"
          " | fp32 L2 {:.2e} max {:.2e}".format(T, *norms(a, want), *norms(b, want), *norms(c, want), *norms(d, want)))
