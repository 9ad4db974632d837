"""Weight-gradient launches of the UNet default config's large layers at batch 8 (the training benchmark's shapes), timed
alone: us per call (kernel + reduction) and TFLOP/s.  With UNETMI_LIBRARY pointing at a -DWG_ABLATE=<bits> build of the
library (csrc/wgrad.hip: 1 no global loads, 2 no LDS stores in the loop, 4 no MFMAs, 8 no step barrier) the same
launches price the stages; such builds give wrong numbers by construction.

    python tools/wgrad_probe.py [batch]"""
import sys

import torch

sys.path.insert(0, ".")
from unet_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
#        name,  C0, C1,  K, kernel, stride, pad, out_pad, transposed, H, W (the layer's input)
CASES = [("enc_1 1x7", 32, 0, 64, (1, 7), (1, 1), (0, 3), (0, 0), 0, 257, 509),
         ("enc_2 7x5 s22", 64, 0, 64, (7, 5), (2, 2), (3, 2), (0, 0), 0, 257, 509),
         ("enc_3 7x5 s21", 64, 0, 64, (7, 5), (2, 1), (3, 2), (0, 0), 0, 129, 255),
         ("dec_6 7x5 s21 T", 64, 64, 64, (7, 5), (2, 1), (3, 2), (0, 0), 1, 65, 255),
         ("dec_7 7x5 s22 T", 64, 64, 64, (7, 5), (2, 2), (3, 2), (0, 0), 1, 129, 255),
         ("dec_8 1x7 T", 64, 64, 64, (1, 7), (1, 1), (0, 3), (0, 0), 1, 257, 509)]
for name, C0, C1, K, k, s, p, op, tr, H, W in CASES:
    geom = [K, k[0], k[1], s[0], s[1], p[0], p[1], 1, 1, op[0], op[1]]
    d = ops.make_desc(N, C0, C1, H, W, geom, [tr, 1 if C1 else 0, 0, 0])
    in0 = torch.randn(N, C0, H, W, device=dev)
    in1 = torch.randn(N, C1, H, W, device=dev) if C1 else None
    dout = torch.randn(N, K, d.OH, d.OW, device=dev)
    flags = [tr, 1 if C1 else 0]
    for _ in range(3):
        ops.conv2d_wgrad(in0, in1, dout, geom, flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv2d_wgrad(in0, in1, dout, geom, flags)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    px = (H * W) if tr else (d.OH * d.OW)
    flop = 2.0 * N * (C0 + C1) * K * k[0] * k[1] * px
    print(f"{name:18s} {us:8.1f} us  {flop / us / 1e6:6.1f} TFLOP/s = {flop / us / 1e6 / 157.3:.3f}", flush=True)
