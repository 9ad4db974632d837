"""Single layers of the UNet default config on channel-blocked fp16 tensors at batch 64: launch time against the row
width (rows of a blocked tensor are 16*W bytes: 128-byte aligned only when W is a multiple of 8; the network's own
T = 4k+1 never is).

    python tools/unet_bf16_width.py [batch]"""
import sys

import torch

sys.path.insert(0, ".")
from unet_amd.UNet import default_unet_blocks  # noqa: E402
from unet_amd.blocked import BlockedConv, BlockedTensor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc, dec = default_unet_blocks()
for m in enc + dec:
    m.to(dev).eval()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


cases = [("enc_0 2->32 7x1", BlockedConv(enc[0].conv, enc[0].bn, enc[0].acti, halves=2, out_fmt="fp16"), 2, 0, "bf16"),
         ("enc_1 32->64 1x7", BlockedConv(enc[1].conv, enc[1].bn, enc[1].acti, out_fmt="fp16"), 32, 0, "fp16"),
         ("dec_8 64+64->64 1x7", BlockedConv(dec[8].transconv, dec[8].bn, dec[8].acti, out_fmt="fp16"), 64, 64, "fp16"),
         ("dec_9 64+32->1 7x1", BlockedConv(dec[9].transconv, dec[9].bn, dec[9].acti, out_fmt="fp16"), 64, 32, "fp16"),
         ("dec_9 to fp32 NCHW", BlockedConv(dec[9].transconv, dec[9].bn, dec[9].acti, nchw_out=True), 64, 32, "fp16")]
with torch.no_grad():
    for name, layer, c0, c1, fmt in cases:
        line = f"{name:22s}"
        for W in (509, 512, 496, 509):
            x = BlockedTensor.from_nchw(torch.randn(B, c0, 257, W, device=dev), fmt=fmt)
            s = BlockedTensor.from_nchw(torch.randn(B, c1, 257, W, device=dev), fmt=fmt) if c1 else None
            us = timed(lambda: layer(x, s))
            line += f"  W={W}: {us:7.1f} us ({us * 1e6 / (B * 257 * W):6.2f} ps/px)"
        print(line, flush=True)
