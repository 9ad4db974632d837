"""Times thin layers (few output channels from many inputs) at full resolution: UNet20's 64 -> 32 1x1 Softplus head,
its first encoder 14 -> 32 (7x1) and the UNet's last decoder 96 -> 1 (7x1) and first encoder 2 -> 32."""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from unet_amd.fused import FusedConv
    dev = torch.device("cuda", 0)
    N, H, W = 32, 257, int(os.environ.get("PROBE_W", "509"))
    cases = [("head 64->32 1x1 softplus", nn.Conv2d(64, 32, 1), None, nn.Softplus(), 64),
             ("enc_0 14->32 7x1", nn.Conv2d(14, 32, (7, 1), 1, (3, 0)), nn.BatchNorm2d(32), nn.LeakyReLU(), 14),
             ("enc_0 2->32 7x1", nn.Conv2d(2, 32, (7, 1), 1, (3, 0)), nn.BatchNorm2d(32), nn.LeakyReLU(), 2),
             ("dec_9 96->1 7x1", nn.ConvTranspose2d(96, 1, (7, 1), 1, (3, 0)), nn.BatchNorm2d(1), nn.LeakyReLU(), 96)]
    for name, conv, bn, act, cin in cases:
        mods = nn.ModuleList([m for m in (conv, bn, act) if m is not None]).to(dev).eval()
        f = FusedConv(conv, bn, act)
        x = torch.randn(N, cin, H, W, device=dev)
        for _ in range(3):
            y = f(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = f(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        gb = (x.numel() + y.numel()) * 4 / 1e9
        print("{:28s} {:8.1f} us  {:5.2f} TB/s algorithmic".format(name, dt * 1e6, gb / dt / 1e3), flush=True)


if __name__ == "__main__":
    main()
