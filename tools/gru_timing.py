"""Time the GRU bottlenecks with the persistent recurrence kernel and with torch's device GRU."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import unet_amd
from unet_amd import gru as G


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    out = {}
    for n_in, hidden, B, T in ((128, 128, 64, 125), (256, 512, 64, 125), (256, 512, 8, 125), (256, 256, 32, 125)):
        gru = torch.nn.GRU(n_in, hidden, batch_first=True).to(dev).eval()
        x = torch.randn(B, T, n_in, device=dev)
        with torch.no_grad():
            a = timed(lambda: G.gru_forward(gru, x))
            b = timed(lambda: gru(x))
        out["gru_{}_{}_B{}_T{}".format(n_in, hidden, B, T)] = {"kernel_ms": round(a, 3), "torch_ms": round(b, 3)}
    for name, build, shape in (("dsunet_tgru", lambda: unet_amd.DSUNet(bottleneck="TGRU"), None),):
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
