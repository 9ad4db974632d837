"""Headline benchmark: audio-seconds processed per second, UNet forward fp32.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no WORLD_SIZE in the environment this
process only starts ``python -m torch.distributed.run --nproc-per-node N`` on
itself as a child (it has not touched the GPU) and exits with the child's
code; under torch.distributed.run each rank binds to its GPU and joins an
RCCL process group.  A step = one forward of the "UNet default config" (the 10+10
layer table of cRFUNet.py:192-218 with c_in=2 and a 1-channel last decoder
around UNet.py's shell) over a batch of 64 synthetic 4-second utterances per
GPU: complex STFT planes N(0,1), n_fft 512 / hop 128 -> F=257, T=501, zero
padded to T=509 (the table only accepts T = 13 mod 16); an utterance counts
4.0 s.  Inputs are resident in HBM before the timed region.  The batch shards
over ranks with no collective in the data path (eval-mode forward has no
cross-sample op); a barrier brackets the timed region and the maximum time
over ranks is used.  Utterance i of the global batch is generated from seed
1234 + i and rank r takes the rows unet_amd.sharding.shard_batch gives it:
by default the global batch is 64 x N (weak scaling), ``--global-batch G``
fixes it (strong scaling).  The line carries ``ranks_seen`` (an all-reduce of
ones over the process group, on the data device) and every rank's own time.

Rank 0 prints ONE JSON line.  "roofline" is for the dominant kernel
(the conv_dma_kernel instantiation with the largest share of the step), from HIP events recorded on the launch stream around each of its
launches inside the timed region: algorithmic FLOPs (2 x MACs of the layer)
/ launch time, against the dense fp32 MFMA peak.  "cpu_baseline" times the
CPU oracle (this repository's restatement of the reference, stock torch CPU
ops) on a batch of 4 on the host cores, rank 0 at N=1 only.
"""

import argparse
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PER_GPU_BATCH = 64
AUDIO_SECONDS = 4.0
F_BINS, T_FRAMES, T_PADDED = 257, 501, 509
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)


def load_launcher():
    """unet_amd/launch.py by path: importing the package would load the HIP library, which the launching parent
    of a multi-rank run must not do."""
    spec = importlib.util.spec_from_file_location("unetmi_launch", os.path.join(ROOT, "unet_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_batch(batch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((batch, 2, F_BINS, T_FRAMES), generator=g)
    return torch.nn.functional.pad(x, (0, T_PADDED - T_FRAMES)).contiguous()


def synthetic_utterances(lo, hi):
    """Rows [lo, hi) of the global batch: utterance i is a function of i alone, so any sharding of the batch
    sees the same utterances."""
    if hi <= lo:
        return torch.zeros((0, 2, F_BINS, T_PADDED))
    return torch.cat([synthetic_batch(1, 1234 + i) for i in range(lo, hi)])


def shard_of(args, per_gpu, ranks):
    """(global batch, lo, hi, "weak" | "strong") for this rank."""
    spec = importlib.util.spec_from_file_location("unetmi_sharding", os.path.join(ROOT, "unet_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    total = args.global_batch if args.global_batch else per_gpu * ranks.world
    lo, hi = sharding.shard_batch(total, ranks.rank, ranks.world)
    return total, lo, hi, ("strong" if args.global_batch else "weak")


def conv_macs(d):
    """Multiply-accumulates of one fused conv launch (ConvDesc)."""
    cin = d.C0 + (d.C1 if d.in_mode == 1 else 0)
    if d.transposed:
        return d.N * d.H * d.W * cin * d.K * d.kh * d.kw
    return d.N * d.OH * d.OW * d.K * cin * d.kh * d.kw


def conv_bytes(d):
    """Algorithmic HBM bytes of one fused conv launch: every input element read once, every output element
    written once (fp32), weights and per-channel parameters neglected."""
    cin = d.C0 + (d.C1 if d.in_mode == 1 else 0)
    return 4 * d.N * (cin * d.H * d.W + d.K * d.OH * d.OW)


def kernel_variant(d):
    """Name of the kernel instantiation that serves this layer, as rocprofv3 prints it: asked of the library
    (unetmi_conv2d_kernel_name), whose plan picks the instantiation from the geometry AND the grid size."""
    from unet_amd.ops import conv_kernel_name
    return conv_kernel_name(d)


def measured_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/summarise_profiles.py), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")


def cpu_baseline(threads):
    import oracle
    from unet_amd import seeded_fill_
    torch.set_num_threads(threads)
    net = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32)).eval()
    seeded_fill_(net, seed=0)
    x = synthetic_batch(4, 1234)
    best = float("inf")
    with torch.no_grad():
        for i in range(2 + 5):
            t0 = time.perf_counter()
            net(x)
            dt = time.perf_counter() - t0
            if i >= 2:
                best = min(best, dt)
    return dict(value=4 * AUDIO_SECONDS / best, unit="audio-s/s", cores=threads, kind="port",
                sample="oracle UNet (same table and weights) forward, batch 4 x [2,257,509] fp32, "
                       "best of 5 after 2 warm-ups, torch {} CPU".format(torch.__version__))


def other_configs(dev, steps=5, warmup=2, batch=32):
    """The other benchmark workloads on this GPU, a few steps each with inputs resident (N=1, after the headline
    measurement; they do not enter value / config / roofline): UNet20 at T=509 + ratio-filter application,
    ResUNet and DSUNet in fp32, the headline's UNet, UNet20, ResUNet and DSUNet on bfloat16 inputs (batch 32 = one
    GPU's share) and the training step at batch 8.  ``frac`` prices the whole step against the peak named by ``bound`` (tools/bench_configs.py)."""
    spec = importlib.util.spec_from_file_location("unetmi_bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
    cfg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfg)
    out = []

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    with torch.no_grad():
        for dtype, names in (("f32", ["unet20", "resunet", "dsunet"]), ("bf16", ["unet", "unet20", "resunet", "dsunet"])):
            g = torch.Generator().manual_seed(4321)
            for w in cfg.iter_workloads(names, batch, dev, dtype, g):
                sec = timed(w["fn"])
                tflops, roof = cfg.roofline_entry(sec, batch, w["gflop"], w["bf16_operands"], w["hbm_gb"])
                out.append({"workload": w["workload"], "name": w["name"], "dtype": dtype, "batch": batch,
                            "ms_per_step": 1e3 * sec, "audio_s_per_s": batch * AUDIO_SECONDS / sec,
                            "bound": roof["bound"], "frac": roof["frac"], "achieved": roof["achieved"],
                            "peak": roof["peak"], "unit": roof["unit"]})
    import unet_amd
    from unet_amd.train import FlatAdam, train_step
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)
    net.train().to(dev)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    x = synthetic_utterances(0, TRAIN_PER_GPU_BATCH).to(dev)
    target = torch.cat([torch.rand((1, 1, F_BINS, T_PADDED), generator=torch.Generator().manual_seed(99 + i))
                        for i in range(TRAIN_PER_GPU_BATCH)]).to(dev)
    sec = timed(lambda: train_step(net, opt, x, target))
    tflops = 3 * 64.34e9 * TRAIN_PER_GPU_BATCH / sec / 1e12
    out.append({"workload": "UNet default config train() step (fwd, MSE, bwd, Adam), batch 8 x [2,257,509]",
                "name": "train", "dtype": "f32", "batch": TRAIN_PER_GPU_BATCH, "ms_per_step": 1e3 * sec,
                "audio_s_per_s": TRAIN_PER_GPU_BATCH * AUDIO_SECONDS / sec, "bound": "mfma",
                "frac": tflops / FP32_MFMA_PEAK_TFLOPS, "achieved": tflops, "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s"})
    return out


TRAIN_PER_GPU_BATCH = 8


def cpu_train_baseline(threads):
    import oracle
    from unet_amd import seeded_fill_
    torch.set_num_threads(threads)
    net = oracle.UNet(*oracle.default_unet(c_in=2, c_last=1, f_dim=32)).train()
    seeded_fill_(net, seed=0)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = synthetic_batch(4, 1234)
    target = torch.rand((4, 1, F_BINS, T_PADDED), generator=torch.Generator().manual_seed(99))
    best = float("inf")
    for i in range(1 + 3):
        t0 = time.perf_counter()
        opt.zero_grad()
        torch.mean((net(x) - target) ** 2).backward()
        opt.step()
        dt = time.perf_counter() - t0
        if i >= 1:
            best = min(best, dt)
    return dict(value=4 * AUDIO_SECONDS / best, unit="audio-s/s", cores=threads, kind="port",
                sample="oracle UNet train step (fwd, MSE, bwd, Adam), batch 4 x [2,257,509] fp32, best of 3 "
                       "after 1 warm-up, torch {} CPU autograd".format(torch.__version__))


def run_train(args, ranks):
    """Data-parallel training step of the same UNet: train-mode forward, MSE to a U(0,1) target mask,
    backward, ONE all-reduce of the flat gradient bucket over RCCL, Adam.  8 utterances per GPU."""
    import unet_amd
    from unet_amd.train import FlatAdam, train_step
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)
    net.train().to(dev)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    total, lo, hi, scaling = shard_of(args, TRAIN_PER_GPU_BATCH, ranks)
    if hi - lo < 1:
        raise SystemExit("train step: rank {} has no utterance (global batch {})".format(rank, total))
    x = synthetic_utterances(lo, hi).to(dev)
    target = torch.cat([torch.rand((1, 1, F_BINS, T_PADDED), generator=torch.Generator().manual_seed(99 + i))
                        for i in range(lo, hi)]).to(dev)
    verbose = os.environ.get("UNETMI_BENCH_VERBOSE") == "1"
    exchange = "torch.distributed ({})".format(ranks.backend)
    if args.comm == "unetmi" and ranks.backend != "gloo":
        # the library's own RCCL communicator (unetmi_comm_init / unetmi_allreduce_f32); its id travels through
        # the torch.distributed group that the launcher set up
        from unet_amd import comm
        comm.init_from_torch_distributed(dev)
        exchange = "unetmi_allreduce_f32 (RCCL, {} ranks)".format(comm.world())
    for i in range(args.warmup):
        loss = train_step(net, opt, x, target)
        if verbose:
            torch.cuda.synchronize()
            print("[rank {}] warm-up step {} loss {:.5f}".format(rank, i, loss.item()), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(net, opt, x, target)
    torch.cuda.synchronize()
    ranks.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = ranks.gather_float(elapsed)
    elapsed = max(per_rank)
    seen = ranks.ranks_seen()
    if rank != 0:
        return
    flops = 3 * 64.34e9 * (hi - lo) * args.steps        # fwd + input-gradient + weight-gradient convs, this GPU
    achieved = flops / elapsed / 1e12
    result = {
        "metric": "audio-seconds trained/sec (whole node), UNet train step fp32",
        "value": total * AUDIO_SECONDS * args.steps / elapsed, "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "UNet default config, train() step: forward with batch statistics, "
                               "mean((mask-target)^2), backward, all-reduce(mean) of one flat gradient bucket, "
                               "Adam(1e-3); batch 8 x [2,257,509] fp32 per GPU",
                   "per_gpu_batch": hi - lo, "global_batch": total,
                   "gflop_per_utterance": 3 * 64.34, "parallelism": "data-parallel x{}".format(world),
                   "final_loss": float(loss.item())},
        "ranks_seen": seen, "backend": ranks.backend, "gradient_exchange": exchange,
        "per_rank_ms": [1e3 * t / args.steps for t in per_rank],
        "roofline": {"bound": "mfma", "kernel": "whole step (conv fwd + dgrad + wgrad)", "achieved": achieved,
                     "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": None},
    }
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_train_baseline(torch.get_num_threads())
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short timings of the other benchmark workloads that an N=1 forward run appends")
    ap.add_argument("--layers", action="store_true", help="per-layer times of the last step on stderr")
    ap.add_argument("--mode", choices=["forward", "train"], default="forward",
                    help="forward: the headline metric (default); train: the data-parallel training step")
    ap.add_argument("--comm", choices=["torch", "unetmi"], default="torch",
                    help="train mode: gradient all-reduce through torch.distributed (default) or through the "
                         "library's own RCCL entry points")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed global batch sharded over the ranks (strong scaling); default 64 per GPU (8 for train)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="host-side check of the launch, sharding and reductions: no GPU, no compute, no measurement")
    args = ap.parse_args()

    launcher = load_launcher()
    if launcher.needs_self_launch(args.gpus):
        # started as one process: the ranks are children of this one, which has not touched the GPU
        sys.exit(launcher.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    if os.environ.get("UNETMI_BENCH_STALL_DUMP"):
        # diagnosis aid: after that many seconds every rank prints its Python stacks and exits
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["UNETMI_BENCH_STALL_DUMP"]), exit=True)
    ranks = launcher.init_ranks(args.gpus, use_gpu=not args.plumbing_only)
    rank, world, dev = ranks.rank, ranks.world, ranks.device

    if args.plumbing_only:
        total, lo, hi, scaling = shard_of(args, PER_GPU_BATCH, ranks)
        ranks.barrier()
        t0 = time.perf_counter()
        ranks.barrier()
        per_rank = ranks.gather_float(time.perf_counter() - t0)
        sizes = ranks.gather_float(hi - lo)
        seen = ranks.ranks_seen()
        if rank == 0:
            print(json.dumps({"plumbing_only": True, "value": None, "n_gpus": world, "scaling": scaling,
                              "ranks_seen": seen, "backend": ranks.backend, "global_batch": total,
                              "per_rank_batch": [int(v) for v in sizes], "per_rank_ms": [1e3 * t for t in per_rank]}),
                  flush=True)
        ranks.shutdown()
        return

    if args.mode == "train":
        run_train(args, ranks)
        ranks.shutdown()
        return

    import unet_amd
    from unet_amd import ops

    net = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=2, c_last=1, f_dim=32))
    unet_amd.seeded_fill_(net, seed=0)       # every rank regenerates the same weights
    net.eval().to(dev)
    total, lo, hi, scaling = shard_of(args, PER_GPU_BATCH, ranks)
    x = synthetic_utterances(lo, hi).to(dev)
    barrier = ranks.barrier

    with torch.no_grad():
        for _ in range(args.warmup):
            y = net(x)
        torch.cuda.synchronize()

        records = []
        shapes = []
        ops.set_launch_observer(lambda name, d, e0, e1: (records.append(
            (kernel_variant(d), conv_macs(d), e0, e1, conv_bytes(d))), shapes.append(
            "{}{}+{}->{} k{}x{} s{}x{} out {}x{}".format("T" if d.transposed else "C", d.C0, d.C1, d.K,
                                                      d.kh, d.kw, d.sh, d.sw, d.OH, d.OW))))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        ops.set_launch_observer(None)
    assert tuple(y.shape) == (hi - lo, 1, F_BINS, T_PADDED)

    per_rank = ranks.gather_float(elapsed)
    elapsed = max(per_rank)
    seen = ranks.ranks_seen()

    if rank == 0 and args.layers:
        per_step = len(records) // args.steps
        for (variant, macs, e0, e1, _), shape in zip(records[-per_step:], shapes[-per_step:]):
            ms = e0.elapsed_time(e1)
            print("{:42s} {:34s} {:8.3f} ms {:7.1f} TFLOP/s".format(
                shape, variant, ms, 2e-9 * macs / ms), file=sys.stderr)
    if rank == 0:
        per_variant = {}
        for variant, macs, e0, e1, nbytes in records:
            acc = per_variant.setdefault(variant, [0.0, 0.0, 0, 0.0])
            acc[0] += 2.0 * macs
            acc[1] += e0.elapsed_time(e1) * 1e-3
            acc[2] += 1
            acc[3] += nbytes
        dominant = max(per_variant, key=lambda k: per_variant[k][1])
        flops, secs, launches, algo_bytes = per_variant[dominant]
        achieved = flops / secs / 1e12
        result = {
            "metric": "audio-seconds processed/sec (whole node), UNet fwd fp32",
            "value": total * AUDIO_SECONDS * args.steps / elapsed,
            "unit": "audio-s/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "ranks_seen": seen, "backend": ranks.backend,
            "per_rank_ms": [1e3 * t / args.steps for t in per_rank],
            "config": {"workload": "UNet default config (cRFUNet.UNet20 layer table, c_in=2, c_last=1) "
                                   "eval forward, batch 64 x [2,257,509] fp32 per GPU "
                                   "(4 s @ 16 kHz, n_fft 512, hop 128, T 501 zero-padded to 509)",
                       "per_gpu_batch": hi - lo, "global_batch": total,
                       "gflop_per_utterance": 64.34, "parallelism": "batch-sharded x{}".format(world)},
            "roofline": {"bound": "mfma", "kernel": dominant, "achieved": achieved,
                         "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": measured_traffic(dominant),
                         "launches": launches, "avg_launch_ms": 1e3 * secs / launches,
                         "gflop_per_launch": flops / launches / 1e9,
                         "algorithmic_bytes_per_launch": algo_bytes / launches,
                         "all_kernels": {k: {"tflops": v[0] / v[1] / 1e12, "ms": 1e3 * v[1] / args.steps,
                                             "launches_per_step": v[2] // args.steps}
                                         for k, v in per_variant.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(torch.get_num_threads())
        if world == 1 and not args.no_other_configs:
            del net, x, y
            torch.cuda.empty_cache()
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # the reference's DSUNet constructor prints; stdout carries ONE line
                result["other_configs"] = other_configs(dev)
        print(json.dumps(result), flush=True)

    ranks.shutdown()


if __name__ == "__main__":
    main()
