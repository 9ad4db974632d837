"""Times single layers of the bf16-operand blocked convolution (DSUNet shapes, batch 32).

    python tools/b16_probe.py [layer ...]        a name may carry "x2" (two weight halves) and / or "+h" (with the
                                                 fused 1x1 two-channel tanh head), e.g. dec9x2+h

Used with build-time experiment variants of the library (make CXXFLAGS+=-DB16_ABLATE=n) to see where a
layer's time goes: 1 = no MFMA loop, 2 = no operand DMA after the prologue, 4 = no epilogue.
"""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LAYERS = {
    # name: (C0, C1, K, kernel, stride, padding, output_padding, transposed, H, W)
    "enc1": (8, 0, 45, (7, 1), (1, 1), (3, 0), (0, 0), False, 257, 512),
    "enc2": (45, 0, 45, (1, 7), (1, 1), (0, 3), (0, 0), False, 257, 512),
    "enc3": (45, 0, 90, (7, 5), (2, 2), (3, 2), (0, 0), False, 257, 512),
    "enc4": (90, 0, 90, (7, 5), (2, 1), (3, 2), (0, 0), False, 129, 256),
    "dec6": (90, 90, 90, (7, 5), (2, 1), (3, 2), (0, 0), True, 65, 256),
    "dec7": (90, 90, 90, (7, 5), (2, 2), (3, 2), (0, 1), True, 129, 256),
    "dec8": (90, 45, 90, (1, 7), (1, 1), (0, 3), (0, 0), True, 257, 512),
    "dec9": (90, 45, 90, (7, 1), (1, 1), (3, 0), (0, 0), True, 257, 512),
    "head": (90, 0, 2, (1, 1), (1, 1), (0, 0), (0, 0), False, 257, 512),
    "enc5": (90, 0, 90, (5, 3), (2, 2), (2, 1), (0, 0), False, 65, 256),
    "enc6": (90, 0, 90, (5, 3), (2, 1), (2, 1), (0, 0), False, 33, 128),
    "dec4": (90, 90, 90, (5, 3), (2, 1), (2, 1), (0, 0), True, 17, 128),
    "dec5": (90, 90, 90, (5, 3), (2, 2), (2, 1), (0, 0), True, 33, 128),
}


def main():
    from unet_amd.blocked import BlockedConv, BlockedTensor
    names = sys.argv[1:] or list(LAYERS)
    dev = torch.device("cuda", 0)
    N = 32
    from unet_amd import _lib
    for name in names:
        with_head = name.endswith("+h")
        base = name[:-2] if with_head else name
        halves = 2 if base.endswith("x2") else 1
        base = base[:-2] if halves == 2 else base
        C0, C1, K, k, s, p, op, tr, H, W = LAYERS[base]
        conv = (nn.ConvTranspose2d(C0 + C1, K, k, stride=s, padding=p, output_padding=op) if tr
                else nn.Conv2d(C0 + C1, K, k, stride=s, padding=p))
        bn = nn.BatchNorm2d(K).eval()
        block = BlockedConv(conv, bn, nn.LeakyReLU(0.01), halves=halves, nchw_out=(base == "head"))
        head_conv = nn.Conv2d(K, 2, 1).to(dev)
        layer = (lambda a, b: block.with_head(a, b, head_conv, _lib.ACT_TANH)) if with_head else block
        x0 = BlockedTensor(torch.randn(N, 2 * ((C0 + 15) // 16), H, W, 8, device=dev).to(torch.bfloat16), C0)
        x1 = BlockedTensor(torch.randn(N, 2 * ((C1 + 15) // 16), H, W, 8, device=dev).to(torch.bfloat16), C1) if C1 else None
        try:
            for _ in range(3):
                y = layer(x0, x1)
        except Exception as e:                      # a forced mode (UNETMI_B16_MODE) that does not fit this layer
            print("{:5s} not available: {}".format(name, str(e)[-60:]), flush=True)
            continue
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = layer(x0, x1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        oh, ow = (y.shape[1], y.shape[2]) if with_head else (y.shape[2], y.shape[3])
        taps = k[0] * k[1] / ((s[0] * s[1]) if tr else 1)
        flop = 2.0 * (C0 + C1) * K * taps * oh * ow * N
        print("{:9s} {:8.1f} us  {:7.1f} TFLOP/s".format(name, dt * 1e6, halves * flop / dt / 1e12), flush=True)


if __name__ == "__main__":
    main()
