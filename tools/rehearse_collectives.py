"""Collective rehearsal on a one-GPU box.  argv[1] = "gloo2": two ranks on device 0, gloo, a 7.9 MB
device bucket; "nccl1": one rank, RCCL, the real training step with its bucket all-reduce."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def log(*a):
    print("[{} rank {}]".format(mode, os.environ.get("RANK", "0")), *a, flush=True)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if mode == "gloo2":
    dist.init_process_group("gloo")
    for n in (1 << 10, 1 << 17, 1 << 19, 1980997):
        t = torch.ones(n, device=dev)
        t0 = time.time(); dist.all_reduce(t); torch.cuda.synchronize()
        log("all_reduce of", n, "floats ok in %.3f s" % (time.time() - t0), t[0].item())
else:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import unet_amd
    from unet_amd.train import FlatAdam, train_step
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)
    net.train().to(dev)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    x = torch.randn(2, 2, 257, 29, device=dev); t = torch.rand(2, 1, 257, 29, device=dev)
    for i in range(2):
        loss = train_step(net, opt, x, t); torch.cuda.synchronize(); log("step", i, loss.item())
    big = torch.ones(1980997, device=dev); dist.all_reduce(big); torch.cuda.synchronize(); log("rccl all_reduce ok", big[0].item())
dist.destroy_process_group()
log("done")
