"""Per-launch table of one forward of a workload: kernel, layer shape, time, TFLOP/s and fraction of the fp32 MFMA peak
(events recorded around every convolution launch through ops.set_launch_observer).

    python tools/layer_table.py [unet20|resunet|dsunet] [batch] [f32|bf16]"""
import sys

import torch

sys.path.insert(0, ".")
import contextlib  # noqa: E402

from unet_amd import ops  # noqa: E402
from tools.bench_configs import iter_workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dsunet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
DT = sys.argv[3] if len(sys.argv) > 3 else "f32"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
with contextlib.redirect_stdout(sys.stderr):
    gen = iter_workloads([name], B, dev, DT, g)
    wl = next(gen)
    step = wl["fn"]
    with torch.no_grad():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        rec = []
        ops.set_launch_observer(lambda op, d, e0, e1: rec.append((op, d, e0, e1)))
        step()
        torch.cuda.synchronize()
        ops.set_launch_observer(None)
tot = 0.0
for op, d, e0, e1 in rec:
    ms = e0.elapsed_time(e1)
    C = d.C0 + d.C1 if d.in_mode == 1 else d.C0
    if d.transposed:
        macs = d.N * C * d.K * d.kh * d.kw * d.H * d.W
    else:
        macs = d.N * C * d.K * d.kh * d.kw * d.OH * d.OW
    tf = 2 * macs / ms / 1e9
    tot += ms
    b16 = op.startswith("conv2d_b16")
    try:
        kn = ops.conv_b16_kernel_name(d, 2 if "head" in op else 0) if b16 else ops.conv_kernel_name(d)
    except Exception:
        kn = op
    if b16 and d.reserved[1] == 2:
        macs *= 2
        kn += " x2"
    print("%-34s %s %3d+%3d->%3d  %3dx%-3d k%dx%d s%dx%d  %8.1f us  %6.1f TF  %.2f" % (
        kn[:34], "T" if d.transposed else "C", d.C0, d.C1, d.K, d.H, d.W, d.kh, d.kw, d.sh, d.sw, ms * 1e3, tf, tf / (2500.0 if b16 else 157.3)))
print("sum of observed launches %.2f ms" % tot)
