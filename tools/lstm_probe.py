"""Times ResUNet's bottleneck LSTM (501 steps, batch 32) and checks it against torch's CPU LSTM."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_amd
from unet_amd.lstm import lstm_forward
dev = torch.device("cuda", 0)
ref = torch.nn.LSTM(384, 640, 3, batch_first=True, proj_size=384)
holder = torch.nn.Module(); holder.lstm = ref
unet_amd.seeded_fill_(holder, seed=71)
lstm = copy.deepcopy(ref).to(dev)
x = torch.randn(32, 501, 384)
with torch.no_grad():
    want = ref(x)[0]
xd = x.to(dev)
cache = {}
for _ in range(2):
    y = lstm_forward(lstm, xd, cache)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    y = lstm_forward(lstm, xd, cache)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
worst = 0.0
for _ in range(20):
    y = lstm_forward(lstm, xd, cache)
    worst = max(worst, ((y.cpu() - want).abs().max() / want.abs().max()).item())
print("%.2f ms per forward, worst rel-err over 20 runs %.2e" % (dt * 1e3, worst), flush=True)
