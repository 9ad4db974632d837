"""Times the fused residual-block layer kernel on the top-level ResUNet tensor (fp32 and bf16 storage)."""
import os, sys, time, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(dtype):
    import unet_amd
    from unet_amd.ops import resblock_layer
    dev = torch.device("cuda")
    x = torch.randn(32, 64, 257, 501, device=dev).to(dtype)
    dw = torch.randn(64, 1, 3, 3, device=dev); db = torch.randn(64, device=dev)
    pw = torch.randn(64, 64, 1, 1, device=dev) * 0.1; pb = torch.randn(64, device=dev)
    al = torch.full((1,), 0.25, device=dev); sc = torch.ones(64, device=dev); sh = torch.zeros(64, device=dev)
    for d in (1, 16):
        for _ in range(2): resblock_layer(x, None, dw, db, pw, pb, al, sc, sh, d, True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): resblock_layer(x, None, dw, db, pw, pb, al, sc, sh, d, True)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        gb = 2 * x.numel() * x.element_size() / 1e9
        print("%s d=%-2d %.3f ms  %.2f TB/s algorithmic" % (str(dtype)[6:], d, ms, gb / ms), flush=True)
if len(sys.argv) > 1:
    run(torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32)
else:
    subprocess.run([sys.executable, __file__, "f32"])
    subprocess.run([sys.executable, __file__, "bf16"])
