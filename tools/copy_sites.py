"""Which Python lines launch copies / small torch kernels inside one eval forward of a model (default: ResUNet)."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import unet_amd


def main():
    dev = torch.device("cuda:0")
    name = sys.argv[1] if len(sys.argv) > 1 else "resunet"
    if name == "resunet":
        net = unet_amd.ResUNet()
        x = torch.randn(2, 4, 257, 501, device=dev)
        args = (x,)
    else:
        net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
        args = (torch.randn(2, 2, 257, 509, device=dev),)
    unet_amd.seeded_fill_(net, seed=0)
    net.eval().to(dev)
    with torch.no_grad():
        for _ in range(2):
            net(*args)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
            net(*args)
            torch.cuda.synchronize()
    rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.count)
    for e in rows[:40]:
        print("%5d %-40s %s" % (e.count, e.key[:40], str(e.input_shapes)[:100]))
    print("---- stacks of copy_/to/_to_copy")
    rows = prof.key_averages(group_by_stack_n=6)
    for e in sorted(rows, key=lambda e: -e.count):
        if any(k in e.key for k in ("copy_", "_to_copy", "aten::to", "contiguous", "clone")) and e.count >= 2:
            print(e.count, e.key, [s for s in e.stack if "unet_amd" in s][:3])


if __name__ == "__main__":
    main()
