import sys, time, torch
sys.path.insert(0, "/root/repo")
import unet_amd
from unet_amd.train import FlatAdam, train_step
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = unet_amd.ResUNet(c_in=4, c_out=1)
unet_amd.seeded_fill_(net, seed=0)
net.train().to(dev)
opt = FlatAdam(net.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1)
x = torch.randn((B, 4, 257, 501), generator=g).to(dev)
target = torch.rand((B, 2, 257, 501), generator=g).to(dev)
for _ in range(2):
    l = train_step(net, opt, x, target)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    l = train_step(net, opt, x, target)
torch.cuda.synchronize()
print("ResUNet train step B=%d: %.1f ms, loss %.5f, peak mem %.1f GB" % (B, (time.perf_counter() - t0) / 3 * 1e3, float(l), torch.cuda.max_memory_allocated() / 2**30))
