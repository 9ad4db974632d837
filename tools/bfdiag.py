import sys, torch
sys.path.insert(0, '.')
import oracle, unet_amd
from unet_amd.seeding import normal
from tests.util import rel_err
dev = torch.device('cuda')
# resblock 64 bf16
ref = oracle.ResBlock(64).eval(); unet_amd.seeded_fill_(ref, seed=33)
m = unet_amd.ResBlock(64); m.load_state_dict(ref.state_dict()); m.eval().to(dev)
x = normal("rb64/x", (2, 64, 9, 70), seed=33).to(torch.bfloat16)
with torch.no_grad():
    want = ref(x.float()); got = m(x.to(dev))
    print("resblock bf16", got.dtype, rel_err(got.float(), want))
    # layer by layer with fp32 truth on the device's own intermediate
    got32 = m(x.float().to(dev)); print("resblock fp32 same input", rel_err(got32, want))
# pwconv64
r = oracle.Encoder(64, 64, (1, 1), (1, 1), activation="PReLU").eval(); unet_amd.seeded_fill_(r, seed=61)
d = unet_amd.Encoder(64, 64, (1, 1), (1, 1), activation="PReLU"); d.load_state_dict(r.state_dict()); d.eval().to(dev)
with torch.no_grad():
    print("pw64 bf16", rel_err(d(x.to(dev)).float(), r(x.float())))
# whole ResUNet: fp32 device on bf16-rounded input, and per stage
ref = oracle.ResUNet().eval(); unet_amd.seeded_fill_(ref, seed=31)
m = unet_amd.ResUNet(); m.load_state_dict(ref.state_dict()); m.eval().to(dev)
xi = normal("resunet/xbf", (2, 4, 257, 24), seed=35).to(torch.bfloat16)
with torch.no_grad():
    want = ref(xi.float())
    print("resunet fp32 dev", rel_err(m(xi.float().to(dev)), want))
    gb = m(xi.to(dev)).float().cpu()
    print("resunet bf16 dev", rel_err(gb, want), "L2-rel", ((gb - want).norm() / want.norm()).item())
    for seed in (36, 37, 38):
        xs = normal("resunet/xbf", (2, 4, 257, 24), seed=seed).to(torch.bfloat16)
        w2 = ref(xs.float()); g2 = m(xs.to(dev)).float().cpu()
        print("  seed", seed, rel_err(g2, w2), "L2-rel", ((g2 - w2).norm() / w2.norm()).item())
    # sensitivity of the reference itself to bf16-level input noise
    noise = xi.float() * (1 + 2**-9 * torch.randn_like(xi.float()))
    print("reference sensitivity to 2^-9 input noise", rel_err(ref(noise), want))
