"""Input gradient of one transposed-conv layer on the device against torch's CPU autograd."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_amd
from unet_amd.train import _input_gradient

torch.manual_seed(0)
for (cin, cout, k, s, p, op, H, W) in ((128, 64, (1, 7), (1, 1), (0, 3), (0, 0), 257, 45),
                                       (128, 64, (7, 5), (2, 2), (3, 2), (0, 0), 129, 23),
                                       (128, 64, (7, 5), (2, 1), (3, 2), (0, 0), 65, 23),
                                       (128, 64, (1, 7), (1, 1), (0, 3), (0, 0), 257, 64)):
    x = torch.randn(2, cin, H, W, requires_grad=True)
    w = torch.randn(cin, cout, *k) * 0.05
    y = torch.nn.functional.conv_transpose2d(x, w, None, s, p, op)
    dz = torch.randn_like(y)
    y.backward(dz)
    geom = [cout, k[0], k[1], s[0], s[1], p[0], p[1], 1, 1, op[0], op[1]]
    dx = _input_gradient(dz.cuda(), w.cuda(), geom, True, (H, W)).cpu()
    err = (dx - x.grad).abs()
    print(k, s, (H, W), "max err {:.2e} of {:.2e}".format(err.max().item(), x.grad.abs().max().item()),
          "worst at", [int(i) for i in torch.nonzero(err == err.max())[0]],
          "bad elements", int((err > 1e-4 * x.grad.abs().max()).sum()))
    bad = (err > 1e-4 * x.grad.abs().max())
    if bad.any():
        idx = torch.nonzero(bad)
        print("   channels", sorted(set(idx[:, 1].tolist()))[:10], "rows", sorted(set(idx[:, 2].tolist()))[:10],
              "cols", sorted(set(idx[:, 3].tolist()))[:20])
