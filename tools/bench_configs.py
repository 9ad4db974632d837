"""Throughput of the secondary workloads on one MI355X (bench.py measures the headline one):

    python tools/bench_configs.py [--steps K] [--warmup W] [--batch B] [--only NAME]

  unet20   cRFUNet.UNet20 forward + ratio-filter application, per-GPU share (32) of the batch of 256
           that is sharded over 8 GPUs, [32,14,257,509] fp32
  resunet  ResUNet forward, [32,4,257,501] fp32
  dsunet   DSUNet (UDSS.set_size(45, 8) table) forward, sf [32,8,257,512], SV [32,4,514], theta [32,2]
  helper   DSUNet_helper, wave in -> wave out: [32,4,64000] samples, STFT and inverse STFT included
  udss     UDSS(type_norm="BatchNorm2d") complex-valued network, sf [32,4,257,512,2], SV [32,4,514], theta [32,2]

One JSON line per workload: audio-seconds/s (4.0 s per utterance), ms per step and the algorithmic
GFLOP per utterance it implies.  ``--gpus N`` starts N ranks (one per GPU, RCCL) like bench.py does: every rank
runs its unet_amd.sharding.shard_batch rows of ``--global-batch`` (default ``--batch`` per GPU; cfg 3 is
``--only unet20 --gpus 8 --global-batch 256``), the step time is the maximum over ranks between barriers and the
value is the whole job's.  Run it under ``rocprofv3 --kernel-trace --stats`` for the per-kernel
times behind the bandwidth figures of the depthwise / mask kernels.
"""

import argparse
import importlib.util
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AUDIO_SECONDS = 4.0


RANKS = None        # unet_amd.launch.Ranks of this process


def load_by_path(name):
    spec = importlib.util.spec_from_file_location("unetmi_" + name, os.path.join(ROOT, "unet_amd", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def timed(fn, steps, warmup):
    """Seconds per step: barrier + synchronize either side of the timed steps, maximum over ranks."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    RANKS.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    RANKS.barrier()
    return max(RANKS.gather_float((time.perf_counter() - t0) / steps))


def roofline_entry(sec, B, gflop, bf16_operands=False, hbm_gb=None):
    """Prices the WHOLE forward against the peak that bounds its dominant kernels: the fp32 or (bfloat16 operands)
    bf16 dense MFMA peak, or -- ResUNet -- HBM bandwidth over the algorithmic bytes of its residual-block layers
    (one read + one write of the activation per layer)."""
    tflops = gflop * B / sec / 1e3
    if hbm_gb is not None:
        ach = hbm_gb * B / sec
        return tflops, {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                        "note": "algorithmic bytes of the ResBlock layers over the whole forward's time"}
    peak = 2500.0 if bf16_operands else 157.3
    return tflops, {"bound": "mfma", "achieved": tflops, "peak": peak, "unit": "TFLOP/s", "frac": tflops / peak}


def iter_workloads(names, B, dev, dtype, g):
    """Yields one dict per workload (name, fn = one step, gflop per utterance, description, dtype, bf16_operands,
    hbm_gb); the model and its inputs are released when the generator is advanced.  ``names``: None = all."""
    import unet_amd
    from unet_amd.ops import crf_apply
    act = torch.bfloat16 if dtype == "bf16" else torch.float32
    want = lambda n: names is None or n in names  # noqa: E731
    if want("unet"):
        net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
        unet_amd.seeded_fill_(net, seed=0)
        net.eval().to(dev)
        x = torch.randn((B, 2, 257, 509), generator=g).to(dev).to(act)
        yield dict(name="unet", fn=lambda: net(x), gflop=64.34, dtype=dtype, bf16_operands=dtype == "bf16", hbm_gb=None,
                   workload="UNet default config forward, [B,2,257,509]" +
                            (" bfloat16 input: 16-bit operands (fp16 behind the first encoder)" if dtype == "bf16" else ""))
        del net, x
    if want("unet20"):
        net = unet_amd.UNet20()
        unet_amd.seeded_fill_(net, seed=21)
        net.T = 509
        net.eval().to(dev)
        x = torch.rand((B, 14, 257, 509), generator=g).to(dev).to(act)
        X = torch.view_as_complex(torch.randn((B, 4, 257, 509, 2), generator=g).contiguous()).to(dev)
        yield dict(name="unet20", fn=lambda: crf_apply(net(x), X), gflop=76.65, dtype=dtype,
                   bf16_operands=dtype == "bf16", hbm_gb=None,
                   workload="cRFUNet.UNet20(c_in=14, c_out=4) forward at T=509 + ratio-filter application, one GPU's "
                            "share of a batch of 256 over 8 GPUs")
        del net, x, X
    if want("resunet"):
        net = unet_amd.ResUNet()
        unet_amd.seeded_fill_(net, seed=31)
        net.eval().to(dev)
        x = torch.randn((B, 4, 257, 501), generator=g).to(dev).to(act)
        yield dict(name="resunet", fn=lambda: net(x), gflop=46.0, dtype=dtype, bf16_operands=False,
                   hbm_gb=1.285,      # a bfloat16 input is widened once: the tensors between the layers are fp32 either way
                   workload="ResUNet() forward, [B,4,257,501]" + (" bfloat16" if dtype == "bf16" else ""))
        del net, x
    if want("dsunet"):
        net = unet_amd.DSUNet(unet_amd.dsunet_architecture(45, 8))
        unet_amd.seeded_fill_(net, seed=41)
        net.eval().to(dev)
        sf = torch.randn((B, 8, 257, 512), generator=g).to(dev).to(act)
        SV = torch.randn((B, 4, 514), generator=g).to(dev)
        ang = torch.rand((B,), generator=g) * 2 * math.pi
        theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1).to(dev)
        yield dict(name="dsunet", fn=lambda: net(sf, SV, theta), gflop=129.76, dtype=dtype,
                   bf16_operands=dtype == "bf16", hbm_gb=None,
                   workload="DSUNet(UDSS table, model_complexity 45, 8 inputs) forward, sf [B,8,257,512], "
                            "SV [B,4,514], theta [B,2]")
        del net, sf, SV, theta
    if want("helper"):
        net = unet_amd.DSUNet_helper(unet_amd.dsunet_architecture(45, 8))
        net.bf16_matrix = dtype == "bf16"
        unet_amd.seeded_fill_(net, seed=51)
        net.eval().to(dev)
        wav = (torch.randn((B, 4, 64000), generator=g) * 0.1).to(dev)
        angle = (torch.rand((B,), generator=g) * 360).to(dev)
        mic = ((torch.rand((B, 4, 3), generator=g) - 0.5) * 0.1).to(dev)
        yield dict(name="helper", fn=lambda: net(wav, angle, mic), gflop=129.76, dtype=dtype,
                   bf16_operands=dtype == "bf16", hbm_gb=None,
                   workload="DSUNet_helper wave -> wave: 4-channel 4 s 16 kHz audio, STFT + features + DSUNet + mask "
                            "+ inverse STFT, [B,4,64000] -> [B,64000]")
        del net, wav, angle, mic
    if want("udss"):
        net = unet_amd.UDSS(type_norm="BatchNorm2d")
        net.bf16_matrix = True                       # with bf16: bfloat16 operands (opt-in for this network)
        unet_amd.seeded_fill_(net, seed=61)
        net.eval().to(dev)
        sf = torch.randn((B, 4, 257, 512, 2), generator=g).to(dev).to(act)
        SV = torch.randn((B, 4, 514), generator=g).to(dev)
        ang = torch.rand((B,), generator=g) * 2 * math.pi
        theta = torch.stack([torch.sin(ang), torch.cos(ang)], dim=1).to(dev)
        yield dict(name="udss", fn=lambda: net(sf, SV, theta), gflop=round(udss_gflop(), 2), dtype=dtype,
                   bf16_operands=dtype == "bf16", hbm_gb=None,
                   workload="UDSS(type_norm='BatchNorm2d') complex-valued forward, sf [B,4,257,512,2], SV [B,4,514], "
                            "theta [B,2]")
        del net, sf, SV, theta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default=None)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="activation storage of the resunet / dsunet workloads (arithmetic is fp32 either way)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--global-batch", type=int, default=0, help="fixed global batch sharded over the ranks")
    args = ap.parse_args()
    global RANKS
    launcher = load_by_path("launch")
    if launcher.needs_self_launch(args.gpus):
        sys.exit(launcher.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    RANKS = launcher.init_ranks(args.gpus)
    from unet_amd.sharding import shard_batch
    dev = RANKS.device
    total = args.global_batch if args.global_batch else args.batch * RANKS.world
    lo, hi = shard_batch(total, RANKS.rank, RANKS.world)
    B = hi - lo
    g = torch.Generator().manual_seed(4321 + lo)
    seen = RANKS.ranks_seen()
    names = None if args.only is None else args.only.split(",")
    with torch.no_grad():
        for w in iter_workloads(names, B, dev, args.dtype, g):
            sec = timed(w["fn"], args.steps, args.warmup)
            if RANKS.rank != 0:
                continue
            tflops, roof = roofline_entry(sec, B, w["gflop"], w["bf16_operands"], w["hbm_gb"])   # per GPU (rank 0's share)
            print(json.dumps({"workload": w["name"], "value": total * AUDIO_SECONDS / sec, "unit": "audio-s/s",
                              "ms_per_step": 1e3 * sec, "batch": B, "global_batch": total, "n_gpus": RANKS.world,
                              "ranks_seen": seen, "scaling": "strong" if args.global_batch else "weak",
                              "dtype": w["dtype"], "gflop_per_utterance": w["gflop"], "tflops": tflops,
                              "roofline": roof, "config": w["workload"]}), flush=True)
    RANKS.shutdown()


def udss_gflop():
    """Algorithmic GFLOP per utterance of UDSS at F=257, T=512: a complex MAC is 4 real MACs."""
    from unet_amd.UDSS import udss_table
    t = udss_table(31, 4)
    H, W, macs = 257, 512, 0
    sizes = []
    for i in range(10):
        k, s, p = t["enc_kernel_sizes"][i], t["enc_strides"][i], t["enc_paddings"][i]
        sizes.append((H, W))
        H, W = (H + 2 * p[0] - k[0]) // s[0] + 1, (W + 2 * p[1] - k[1]) // s[1] + 1
        macs += 4 * t["enc_channels"][i] * t["enc_channels"][i + 1] * k[0] * k[1] * H * W
    for i in range(10):
        k, s = t["dec_kernel_sizes"][i], t["dec_strides"][i]
        cin = t["dec_channels"][i] + t["enc_channels"][10 - i]
        macs += 4 * cin * t["dec_channels"][i + 1] * k[0] * k[1] * H * W      # per INPUT pixel of a transposed conv
        H, W = sizes[9 - i]
    macs += 4 * t["dec_channels"][-1] * 257 * 512
    return 2 * macs / 1e9


if __name__ == "__main__":
    main()
