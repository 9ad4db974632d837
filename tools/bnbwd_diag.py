"""bn_act_bwd on a channel slice of a wider tensor against the same call on a contiguous copy."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd.train import bn_act_bwd, bn_train_stats

torch.manual_seed(0)
for (N, C, Cw, H, W) in ((2, 64, 128, 257, 45), (2, 16, 32, 257, 29), (2, 64, 128, 257, 44), (8, 64, 128, 257, 509)):
    z = torch.randn(N, C, H, W, device="cuda")
    wide = torch.randn(N, Cw, H, W, device="cuda")
    dy = wide[:, :C]
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mean, invstd = bn_train_stats(z, 1e-5, 0.1, rm, rv)
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    ap = torch.tensor([0.01], device="cuda")
    a = bn_act_bwd(dy, z, mean, invstd, g, b, 1, ap, True)
    c = bn_act_bwd(dy.contiguous(), z, mean, invstd, g, b, 1, ap, True)
    # reference in float64 on the host
    zz, dd = z.double().cpu(), dy.double().cpu()
    m, istd = zz.mean((0, 2, 3), keepdim=True), 1.0 / torch.sqrt(zz.var((0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
    xhat = (zz - m) * istd
    pre = xhat * g.double().cpu().view(1, -1, 1, 1) + b.double().cpu().view(1, -1, 1, 1)
    dpre = dd * torch.where(pre >= 0, 1.0, 0.01)
    dbeta = dpre.sum((0, 2, 3))
    print((N, C, Cw, H, W), "slice vs contiguous:", [float((x - y).abs().max()) for x, y in zip(a, c)],
          " dbeta vs float64: slice {:.2e} contiguous {:.2e} of {:.2e}".format(
              float((a[2].double().cpu() - dbeta).abs().max()), float((c[2].double().cpu() - dbeta).abs().max()),
              float(dbeta.abs().max())))
