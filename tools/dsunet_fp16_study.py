"""CPU study: which number formats at the two ends of DSUNet's U keep the bfloat16-operand forward within 1e-2 max-norm.
The fp32 oracle is run layer by layer with the roundings the device path applies -- weights and stored activations
rounded to bfloat16 (fp32 accumulation) -- and chosen layers switched to fp16 operands / fp16 storage or to exact
weights (two bfloat16 halves).  Inputs: the full-length test's (T=512, two samples)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
import oracle
import unet_amd
from unet_amd.seeding import normal, uniform

with contextlib.redirect_stdout(sys.stderr):
    ref = oracle.DSUNet(oracle.dsunet_architecture(45, 8)).eval()
unet_amd.seeded_fill_(ref, seed=44)
sf = torch.cat([normal("dsunet/sf512", (1, 8, 257, 512), seed=44), normal("dsunet/sf512b", (1, 8, 257, 512), seed=45)])
sf = sf.to(torch.bfloat16).float()
SV = torch.cat([normal("dsunet/SV512", (1, 4, 514), seed=44), normal("dsunet/SV512b", (1, 4, 514), seed=45)])
ang = torch.cat([uniform("dsunet/theta512", (1,), 0.0, 6.283185, seed=44), uniform("dsunet/theta512b", (1,), 0.0, 6.283185, seed=45)])
theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1)
L = ref.n_layer


def rnd(t, fmt):
    if fmt == "bf16":
        return t.to(torch.bfloat16).float()
    if fmt == "fp16":
        return t.clamp(-65504.0, 65504.0).to(torch.float16).float()
    return t


def conv_of(block):
    for m in block.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            return m
    raise RuntimeError("no conv")


def run(wfmt, ofmt):
    """wfmt / ofmt: dicts keyed ('e', i) / ('d', i) -> 'bf16' | 'fp16' | None (exact); default 'bf16'."""
    saved = []
    with torch.no_grad():
        for kind, blocks in (("e", ref.encoders), ("d", ref.decoders)):
            for i, b in enumerate(blocks):
                c = conv_of(b)
                saved.append((c, c.weight.detach().clone()))
                c.weight.copy_(rnd(c.weight, wfmt.get((kind, i), "bf16")))
        attract = ref.Attractor(SV, theta)
        x = sf
        skips = []
        for i, enc in enumerate(ref.encoders):
            skips.append(ref.res[i](x) if i != L - 1 else x)
            y = ref.dr(enc(x))
            gate = ref.attractEncoders[i](attract)
            x = rnd(torch.reshape(gate, (*gate.shape, 1, 1)) * y, ofmt.get(("e", i), "bf16"))
        p = ref.BTN(x)
        for i, dec in enumerate(ref.decoders):
            p = dec(p)
            if i == L - 1:
                break
            p = rnd(p, ofmt.get(("d", i), "bf16"))
            p = torch.cat([p, skips[L - 1 - i]], dim=1)
        out = ref.mask(ref.linear(p))
        for c, w in saved:
            c.weight.copy_(w)
    return torch.view_as_real(out)


with torch.no_grad():
    want = torch.view_as_real(ref(sf, SV, theta))
e0, dl, dl1 = ("e", 0), ("d", L - 1), ("d", L - 2)
cases = [
    ("all bf16", {}, {}),
    ("two halves: enc0 + last dec (today)", {e0: None, dl: None}, {}),
    ("last dec fp16 weights", {dl: "fp16"}, {}),
    ("enc0 two halves, last dec fp16 weights", {e0: None, dl: "fp16"}, {}),
    ("+ enc0 / dec8 outputs stored fp16", {e0: None, dl: "fp16"}, {e0: "fp16", dl1: "fp16"}),
    ("+ enc1 fp16 weights too", {e0: None, dl: "fp16", ("e", 1): "fp16"}, {e0: "fp16", dl1: "fp16"}),
    ("enc0 bf16 single, rest as above", {dl: "fp16", ("e", 1): "fp16"}, {e0: "fp16", dl1: "fp16"}),
    ("every layer fp16 weights + fp16 storage", {(k, i): "fp16" for k in "ed" for i in range(L)},
     {(k, i): "fp16" for k in "ed" for i in range(L)}),
    ("the same, enc0 weights bf16 (its input is bf16)", {**{(k, i): "fp16" for k in "ed" for i in range(L)}, e0: "bf16"},
     {(k, i): "fp16" for k in "ed" for i in range(L)}),
    ("the same, enc0 two bf16 halves", {**{(k, i): "fp16" for k in "ed" for i in range(L)}, e0: None},
     {(k, i): "fp16" for k in "ed" for i in range(L)}),
]
if len(sys.argv) > 1:
    cases = cases[-int(sys.argv[1]):]
for label, wf, of in cases:
    got = run(wf, of)
    l2 = ((got - want).norm() / want.norm()).item()
    mx = ((got - want).abs().max() / want.abs().max()).item()
    print("{:48s} L2 {:.2e}  max-norm {:.2e}".format(label, l2, mx), flush=True)
