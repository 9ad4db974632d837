"""ResUNet forward on a bfloat16 input, batch 32: the LSTM's recurrent product with fp16 operands against fp32 ones,
alternating on one box (unet_amd.lstm.FP16_RECURRENCE).

    python tools/lstm_fp16_ab.py"""
import contextlib
import sys

import torch

sys.path.insert(0, ".")
import unet_amd.lstm as L  # noqa: E402
from tools.bench_configs import iter_workloads  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    wl = next(iter_workloads(["resunet"], 32, dev, "bf16", g))
step = wl["fn"]


def timed(n=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for mode in (True, False, True, False):
        L.FP16_RECURRENCE = mode
        print("fp16 recurrence" if mode else "fp32 recurrence", round(timed(), 3), "ms", flush=True)
