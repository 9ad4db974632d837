"""Headline workload with the batch starting and ending in pinned HOST memory: copy in, forward, copy out,
not overlapped.  Prints ms per step beside the resident-input figure (a note for the design document;
bench.py's value never includes the copies)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)
    net = net.eval().to(dev)
    x_host = torch.randn(64, 2, 257, 509).pin_memory()
    y_host = torch.empty(64, 1, 257, 509).pin_memory()
    x_dev = x_host.to(dev)

    def resident():
        with torch.no_grad():
            return net(x_dev)

    def through_host():
        with torch.no_grad():
            y = net(x_host.to(dev, non_blocking=True))
        y_host.copy_(y.reshape(y_host.shape), non_blocking=True)

    for name, fn in (("resident", resident), ("host buffers", through_host)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100.0
        print("{}: {:.2f} ms/step, {:.0f} audio-s/s".format(name, ms, 64 * 4.0 / ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
