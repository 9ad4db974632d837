"""Which Python lines launch the small torch kernels (fill, add, index, cat, copy) inside one UNet train step."""
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import unet_amd
from unet_amd.train import FlatAdam, train_step


def main():
    dev = torch.device("cuda:0")
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)
    net.train().to(dev)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    x = torch.randn(2, 2, 257, 125, device=dev)
    target = torch.rand(2, 1, 257, 125, device=dev)
    for _ in range(2):
        train_step(net, opt, x, target)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        train_step(net, opt, x, target)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count)
    for e in rows[:70]:
        print("%5d %-45s %s" % (e.count, e.key[:45], str(e.input_shapes)[:110]))


if __name__ == "__main__":
    main()
