"""Per-parameter gradient error of the full-width UNet train step (device vs float64 CPU), in forward order."""
import copy
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
import unet_amd
from unet_amd.seeding import normal, uniform

T = int(sys.argv[1]) if len(sys.argv) > 1 else 45
ref = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32))
unet_amd.seeded_fill_(ref, seed=31)
dev = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
dev.load_state_dict(ref.state_dict(), strict=True)
xb = normal("unet_full/xb", (2, 2, 257, T), seed=32)
target = uniform("unet_full/target", (2, 1, 257, T), 0.0, 1.0, seed=32)
ref64 = copy.deepcopy(ref).double().train()
ref.train()
torch.mean((ref(xb) - target) ** 2).backward()
torch.mean((ref64(xb.double()) - target.double()) ** 2).backward()
exact = {k: p.grad for k, p in ref64.named_parameters()}
dev.to("cuda").train()
loss = torch.mean((dev(xb.cuda()) - target.cuda()) ** 2)
loss.backward()
for (k, p), (_, q) in zip(dev.named_parameters(), ref.named_parameters()):
    s = exact[k].abs().max().item()
    print("{:32s} scale {:.2e}  device {:.2e}  cpu-fp32 {:.2e}".format(
        k, s, (p.grad.cpu().double() - exact[k]).abs().max().item() / s, (q.grad.double() - exact[k]).abs().max().item() / s))
