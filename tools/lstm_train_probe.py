"""Time one forward + backward of the projected LSTM (nn.LSTM(384, 640, 3, proj_size=384)) at the benchmark's sequence
length: the autograd node on the persistent kernels (unet_amd.lstm.lstm_forward_train) against torch's device LSTM."""
import copy
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from unet_amd import lstm as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 501
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nn.LSTM(384, 640, 3, batch_first=True, proj_size=384).to(dev)
x = torch.randn(B, T, 384, device=dev)
g = torch.randn(B, T, 384, device=dev)


def run(fn, n=5):
    for _ in range(2):
        xi = x.clone().requires_grad_()
        fn(xi).backward(g)
    torch.cuda.synchronize()
    tf = tb = 0.0
    for _ in range(n):
        xi = x.clone().requires_grad_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = fn(xi)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        y.backward(g)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf += t1 - t0
        tb += t2 - t1
    return tf / n * 1e3, tb / n * 1e3, xi.grad


f1, b1, g1 = run(lambda xi: L.lstm_forward_train(net, xi))
print("persistent kernels: forward {:.2f} ms, backward {:.2f} ms  (B={}, T={})".format(f1, b1, B, T), flush=True)
f0, b0, g0 = run(lambda xi: net(xi)[0], n=2)
print("torch device LSTM:  forward {:.2f} ms, backward {:.2f} ms".format(f0, b0))
print("input gradient difference: {:.2e} of its scale".format(((g1 - g0).abs().max() / g0.abs().max()).item()))
