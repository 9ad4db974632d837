"""Times the five layers of a 64-channel residual block on [32, 64, H, 501] in two launch orders:
layer-major (each layer over the whole batch) and chunk-major (all five layers over a few samples at a
time, so a layer's input was written by the launch just before it and may still sit in the memory-side cache).
Calls the C ABI directly with pointers into preallocated buffers, so nothing is allocated inside the loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import unet_amd  # noqa: F401
    from unet_amd import _lib
    from unet_amd._lib import check

    dev = torch.device("cuda")
    B = 32
    stream = torch.cuda.current_stream().cuda_stream
    dw = torch.randn(64, 1, 3, 3, device=dev)
    db = torch.randn(64, device=dev)
    pw = torch.randn(64, 64, 1, 1, device=dev) * 0.1
    pb = torch.randn(64, device=dev)
    al = torch.full((1,), 0.25, device=dev)
    sc = torch.ones(64, device=dev)
    sh = torch.zeros(64, device=dev)

    def layer(n, H, W, d, src, dst, residual):
        check(_lib.lib.unetmi_resblock_layer_fwd(n, 64, H, W, d, src, None, dw.data_ptr(), db.data_ptr(), pw.data_ptr(),
                                                 pb.data_ptr(), al.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                 int(residual), dst, 0, stream), "resblock")

    for H in (257, 128, 63, 31):
        W = 501
        x = torch.randn(B, 64, H, W, device=dev)
        a = torch.empty_like(x)
        b = torch.empty_like(x)
        sample = 64 * H * W * 4
        px = B * H * W * 5

        def block(chunk):
            for c0 in range(0, B, chunk):
                n = min(chunk, B - c0)
                off = c0 * sample
                src, bufs = x.data_ptr() + off, (a.data_ptr() + off, b.data_ptr() + off)
                for i, d in enumerate((1, 2, 4, 8, 16)):
                    dst = bufs[i & 1]
                    layer(n, H, W, d, src, dst, i > 0)
                    src = dst

        want = None
        for chunk in (32, 16, 8, 4, 2, 1):
            for _ in range(2):
                block(chunk)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                block(chunk)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            got = a.clone()      # layer 4 (even index) lands in a
            if want is None:
                want = got
            same = torch.equal(got, want)
            print("H=%-3d chunk=%-2d %.3f ms  %.1f ps/pixel-layer  %.2f TB/s algorithmic  same=%s" % (
                H, chunk, ms, ms * 1e9 / px, 2 * x.numel() * 4 * 5 / 1e9 / ms, same), flush=True)


if __name__ == "__main__":
    main()
