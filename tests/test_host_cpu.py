"""Host-side logic that needs no GPU: the C-ABI library loads and exports what
include/unetmi.h declares, descriptor validation, weight packing, drop-in
module structure, deterministic seeding and the data-parallel sharding."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from unet_amd import _lib
    header = open(os.path.join(ROOT, "include", "unetmi.h")).read()
    declared = set(re.findall(r"\b(unetmi_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "libunetmi.so does not export " + sym
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert _lib.lib.unetmi_version() >= 100


def test_descriptor_validation_and_error_text():
    from unet_amd import _lib
    from unet_amd.ops import make_desc
    d = make_desc(1, 4, 0, 16, 16, [8, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0], [0, 0, 0, 0])
    assert _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d)) > 0
    d.OH += 1                                   # inconsistent output size
    assert _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d)) == 0
    assert "output size" in _lib.last_error()
    d = make_desc(1, 4, 0, 16, 16, [8, 3, 3, 2, 2, 1, 1, 2, 2, 0, 0], [1, 0, 0, 0])
    assert _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d)) == 0   # stride 2 with dilation 2
    assert "dilation" in _lib.last_error()
    # compute entry point rejects null tensors without touching a device
    d = make_desc(1, 4, 0, 16, 16, [8, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0], [0, 0, 0, 0])
    rc = _lib.lib.unetmi_conv2d_fwd(ctypes.byref(d), *([None] * 10), None)
    assert rc == _lib.E_BAD_ARG


def test_pack_weights_is_a_permutation_with_zero_padding():
    from unet_amd.ops import pack_conv_weights
    for transposed, shape in ((0, (64, 6, 7, 5)), (1, (6, 64, 7, 5))):
        w = torch.arange(1, 1 + int(np.prod(shape)), dtype=torch.float32).reshape(shape)
        geom = [64, 7, 5, 2, 2, 3, 2, 1, 1, 0, 0]
        packed = pack_conv_weights(w, 6, 0, geom, [transposed, 0, 0, 0])
        nz = packed[packed != 0]
        assert nz.numel() == w.numel()
        assert torch.equal(torch.sort(nz).values, torch.sort(w.reshape(-1)).values)
    # thin layer -> direct kernel image [ci][kh][kpad]; transposed layers are flipped along kh
    w = torch.arange(1, 1 + 12 * 7, dtype=torch.float32).reshape(12, 1, 7, 1)
    packed = pack_conv_weights(w, 12, 0, [1, 7, 1, 1, 1, 3, 0, 1, 1, 0, 0], [1, 0, 0, 0])
    assert packed.numel() == 12 * 7
    assert torch.equal(packed.reshape(12, 7), w.reshape(12, 7).flip(1))


def test_dropin_modules_have_reference_structure():
    import oracle
    import unet_amd
    dev = unet_amd.UNet(*unet_amd.default_unet_blocks(f_dim=8))
    ref = oracle.UNet(*oracle.default_unet(f_dim=8))
    assert list(dev.state_dict().keys()) == list(ref.state_dict().keys())
    dev.load_state_dict(ref.state_dict(), strict=True)
    assert dev.len_model == 10 and isinstance(dev.encoders, list) and dev.encoders[-1].conv.out_channels == 16
    for name in ("conv", "bn", "acti", "dropout"):
        assert hasattr(dev.encoders[0], name)
    for name in ("transconv", "bn", "acti"):
        assert hasattr(dev.decoders[0], name)
    with pytest.raises(Exception, match="unmatched legnth"):
        unet_amd.UNet(dev.encoders, dev.decoders[:-1])
    with pytest.raises(Exception, match="unknown normalization"):
        unet_amd.Encoder(2, 4, (3, 3), 1, norm="GroupNorm")
    with pytest.raises(RuntimeError, match="no CPU path"):
        dev.eval()(torch.zeros(1, 2, 257, 13))


def test_seeding_is_deterministic_and_name_keyed():
    from unet_amd.seeding import seeded_fill_, uniform01
    a, b = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)), nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
    seeded_fill_(a, seed=5), seeded_fill_(b, seed=5)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert not np.array_equal(uniform01("x", 8, 1), uniform01("y", 8, 1))
    assert np.array_equal(uniform01("x", 8, 1)[:4], uniform01("x", 4, 1))
    assert (a[1].running_var > 0.49).all()


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import oracle
from unet_amd.sharding import shard_batch
from unet_amd.seeding import seeded_fill_, normal
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
net = oracle.UNet(*oracle.default_unet(f_dim=8)).eval()
seeded_fill_(net, seed=3)                      # every rank regenerates identical weights
x = normal("shard/x", (5, 2, 257, 13), seed=3)  # 5 utterances over 2 ranks: 3 + 2
lo, hi = shard_batch(x.shape[0], rank, world)
with torch.no_grad():
    mine = net(x[lo:hi])
    whole = net(x)
# CPU convolution picks its algorithm per batch size, so rows agree to rounding, not bitwise
assert (mine - whole[lo:hi]).abs().max().item() <= 1e-5 * whole.abs().max().item(), "shard differs from full batch"
counts = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
dist.all_gather(counts, torch.tensor([hi - lo]))
assert sum(int(c) for c in counts) == x.shape[0]
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)        # the bench's max-over-ranks timing reduction
assert t.item() == world
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_batch_sharding_two_ranks_gloo(tmp_path):
    """N-rank result == rows of the 1-rank result (no collective in the data path)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_shard_batch_covers_everything_once():
    from unet_amd.sharding import shard_batch
    for total in (0, 1, 5, 64, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


TRAIN_WORKER = r'''
import os, sys, copy, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import oracle
from unet_amd.train import FlatBucket
from unet_amd.seeding import seeded_fill_, normal, uniform
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
net = oracle.UNet(*oracle.default_unet(f_dim=8)).train()
seeded_fill_(net, seed=5)
twin = copy.deepcopy(net)                      # untouched copy for the single-process answer
x = normal("dp/x", (4, 2, 257, 13), seed=5)
target = uniform("dp/t", (4, 1, 257, 13), 0.0, 1.0, seed=5)
per = x.shape[0] // world
# every rank: its own shard, per-replica BatchNorm statistics, gradients into the flat bucket
bucket = FlatBucket(net.parameters())
assert all(p.data_ptr() >= bucket.flat.data_ptr() for p in net.parameters())
bucket.zero_grad()
torch.mean((net(x[rank * per:(rank + 1) * per]) - target[rank * per:(rank + 1) * per]) ** 2).backward()
n = bucket.all_reduce()
assert n == world
mean_grad = bucket.grad / n
# expected: arithmetic mean over shards of the per-shard gradients, computed serially
want = torch.zeros_like(bucket.grad)
for r in range(world):
    m = copy.deepcopy(twin)
    torch.mean((m(x[r * per:(r + 1) * per]) - target[r * per:(r + 1) * per]) ** 2).backward()
    want += torch.cat([p.grad.reshape(-1) for p in m.parameters()])
want /= world
err = (mean_grad - want).abs().max().item()
assert err <= 1e-6 * want.abs().max().item() + 1e-12, err
# every rank holds the same averaged bucket
gathered = [torch.zeros_like(mean_grad) for _ in range(world)]
dist.all_gather(gathered, mean_grad)
assert all(torch.equal(g, gathered[0]) for g in gathered)
# zero_grad re-attaches views if an optimizer replaced .grad
next(iter(net.parameters())).grad = None
bucket.zero_grad()
assert next(iter(net.parameters())).grad.data_ptr() == bucket.grad.data_ptr()
# detach_grads / collect_grads: gradients gathered after the backward pass are those accumulated into the views
shard = slice(rank * per, (rank + 1) * per)
bucket.zero_grad()
torch.mean((net(x[shard]) - target[shard]) ** 2).backward()
accumulated = bucket.grad.clone()
bucket.detach_grads()
assert all(p.grad is None for p in net.parameters())
torch.mean((net(x[shard]) - target[shard]) ** 2).backward()
bucket.collect_grads()
assert torch.equal(bucket.grad, accumulated)
lo, hi = bucket.grad.data_ptr(), bucket.grad.data_ptr() + 4 * bucket.grad.numel()
assert all(lo <= p.grad.data_ptr() < hi for p in net.parameters())
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gradient_bucket_all_reduce_two_ranks_gloo(tmp_path):
    """Data-parallel training exchange: ONE all-reduce over the flat gradient bucket gives the
    mean of the per-shard gradients (BatchNorm statistics stay per replica)."""
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29519", str(script), ROOT],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def _bf16_bits(x):
    """Nearest-even bfloat16 bit patterns of a float32 array (numpy restatement of the packer's rounding)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return ((u >> 16) & 0xFFFF).astype(np.uint16)


def test_b16_weight_image_layout():
    """unetmi_conv2d_b16_pack_weights (a host function): the image is
    [phase][16-channel group][weight half][tap row][column phase][tap col][32-channel tile][k half][row][8 k],
    channels of the two concat sources padded separately to multiples of 16, transposed layers split into
    stride phases with the taps of each phase only."""
    import torch
    from unet_amd.blocked import pack_b16_weights
    rng = np.random.default_rng(7)
    # plain convolution, concat input 20 + 12 channels, 40 outputs, two weight halves
    C0, C1, K, kh, kw = 20, 12, 40, 3, 2
    w = rng.standard_normal((K, C0 + C1, kh, kw)).astype(np.float32)
    img = pack_b16_weights(torch.from_numpy(w), C0, C1, [K, kh, kw, 1, 1, 1, 0, 1, 1, 0, 0], 0, halves=2).numpy().view(np.uint16)
    G0, G1, MT = 2, 1, 2
    assert img.size == (G0 + G1) * kh * kw * 2 * MT * 2 * 32 * 8
    img = img.reshape(G0 + G1, 2, kh, kw, MT, 2, 32, 8)          # [group][weight half][tap row][tap col][tile][k half][row][k]
    hi = _bf16_bits(w)
    lo = _bf16_bits(w - (hi.astype(np.uint32) << 16).view(np.float32))
    for g, hf, k, co in [(0, 0, 0, 0), (0, 1, 7, 39), (1, 0, 3, 17), (1, 0, 4, 5), (2, 0, 5, 33), (2, 1, 3, 8), (2, 1, 4, 8), (0, 0, 1, 40)]:
        cl = (g if g < G0 else g - G0) * 16 + hf * 8 + k
        ci = (cl if cl < C0 else -1) if g < G0 else (C0 + cl if cl < C1 else -1)
        for th in range(kh):
            for tw in range(kw):
                got = img[g, :, th, tw, co // 32, hf, co % 32, k]
                if ci < 0 or co >= K:
                    assert got[0] == 0 and got[1] == 0
                else:
                    assert got[0] == hi[co, ci, th, tw] and got[1] == lo[co, ci, th, tw]
    # transposed, stride (2,2), kernel 5x3, padding (2,1): row phases {kh0=0: 3 taps, kh0=1: 2 taps}, each holding
    # both column phases (column phase 0 starts at kw0 = 1: one tap; column phase 1 at kw0 = 0: two taps)
    C, K = 16, 32
    wt = rng.standard_normal((C, K, 5, 3)).astype(np.float32)
    img = pack_b16_weights(torch.from_numpy(wt), C, 0, [K, 5, 3, 2, 2, 2, 1, 1, 1, 0, 1], 1).numpy().view(np.uint16)
    hb = _bf16_bits(wt)
    # phase ah=0: kh0 = (0+2)%2 = 0 -> ky 0,2,4 ; phase ah=1: kh0 = 1 -> ky 1,3.  Tap row layout [cp0 taps | cp1 taps].
    n0 = 1 * 3 * 3 * 1 * 1 * 2 * 32 * 8
    ph0 = img[:n0].reshape(3, 3, 2, 32, 8)          # [th][cp0: kx=1 | cp1: kx=0, kx=2][half][row][k]
    ph1 = img[n0:].reshape(2, 3, 2, 32, 8)
    for th, ky in enumerate((0, 2, 4)):
        for slot, kx in enumerate((1, 0, 2)):
            assert np.array_equal(ph0[th, slot, 1, :, 2], hb[10, :, ky, kx])    # channel 8 + 2, all 32 outputs
    for th, ky in enumerate((1, 3)):
        for slot, kx in enumerate((1, 0, 2)):
            assert np.array_equal(ph1[th, slot, 0, :, 5], hb[5, :, ky, kx])


def _json_lines(text):
    import json
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            out.append(json.loads(line))
    return out


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` started as ONE process starts its two ranks itself (torch.distributed.run as a
    child), shards the global batch with shard_batch and rank 0 prints one line carrying ranks_seen and every
    rank's time.  Host-side check (--plumbing-only: gloo, no GPU, no compute)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "2"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["per_rank_batch"] == [64, 64] and len(line["per_rank_ms"]) == 2
    # a fixed global batch (256 utterances over 3 ranks) is strong scaling, rows differ by at most one
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--plumbing-only",
                          "--global-batch", "256"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    line = _json_lines(out.stdout)[0]
    assert line["ranks_seen"] == 3 and line["scaling"] == "strong" and sum(line["per_rank_batch"]) == 256


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr


def test_every_entry_point_is_a_dispatcher_op_with_shape_inference():
    """The training, recurrence and blocked-layout entry points are torch.ops.unetmi.* ops like the forward ones:
    under FakeTensorMode (no device, no kernel) each returns tensors of the shapes and dtypes the HIP path gives."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    import unet_amd  # noqa: F401  (registers the ops)
    ops = torch.ops.unetmi
    geom = [4, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0]
    with FakeTensorMode():
        z, c = torch.empty(2, 8, 5, 7), torch.empty(8)
        mean, invstd = ops.bn_train_stats(z, 1e-5, 0.1, c, c.clone())
        assert mean.shape == invstd.shape == (8,)
        assert ops.bn_act_fwd(z, mean, invstd, c, c, 1, None).shape == z.shape
        dz, dg, db, dbias = ops.bn_act_bwd(z, z, mean, invstd, c, c, 1, None, True)
        assert dz.shape == z.shape and dg.shape == db.shape == dbias.shape == (8,)
        assert ops.conv2d_wgrad(z, None, torch.empty(2, 4, 5, 7), geom, [0, 0]).shape == (4, 8, 3, 3)
        assert ops.conv2d_wgrad(z, None, torch.empty(2, 4, 5, 7), geom, [1, 0]).shape == (8, 4, 3, 3)
        status = ops.lstm_recurrence(torch.empty(2560, 640), torch.empty(5, 2560, 16), torch.empty(6, 640, 16), True, False)
        assert status.shape == () and status.dtype == torch.int32
        status = ops.lstm_recurrence_train(torch.empty(2560, 640), torch.empty(2560, 6, 16), torch.empty(640, 6, 16),
                                           torch.empty(640, 6, 16))
        assert status.shape == () and status.dtype == torch.int32
        status = ops.lstm_bptt(torch.empty(2560, 640), torch.empty(2560, 6, 16), torch.empty(640, 6, 16), torch.empty(640, 6, 16))
        assert status.shape == () and status.dtype == torch.int32
        dwg, dbg = ops.dwconv2d_wgrad(z, None, z, [3, 3], [1, 2], [1, 2])
        assert dwg.shape == (8, 1, 3, 3) and dbg.shape == (8,)
        yb, mb, ib = torch.ops.unetmi.bn_act(z, c, c, None, None, 1, 1e-5, 0, None)
        assert yb.shape == z.shape and mb.shape == ib.shape == (8,)
        status = ops.gru_recurrence(torch.empty(384, 128), torch.empty(384), torch.empty(5, 384, 16), torch.empty(6, 128, 16), False)
        assert status.shape == () and status.dtype == torch.int32
        assert ops.layernorm_w_act(z, None, None, 1e-5, 1, torch.empty(1), None).shape == z.shape
        assert ops.gru_gate(torch.empty(7, 2, 768), 0, 1, None, torch.empty(384), torch.empty(7, 2, 256), -1) is None
        status = ops.gru_recurrence_train(torch.empty(384, 128), torch.empty(384), torch.empty(5, 384, 16), torch.empty(6, 128, 16),
                                          torch.empty(5, 512, 16))
        assert status.shape == () and status.dtype == torch.int32
        dh0, status = ops.gru_bptt(torch.empty(384, 128), torch.empty(5, 512, 16), torch.empty(6, 128, 16), torch.empty(5, 128, 16))
        assert dh0.shape == (128, 16) and status.shape == () and status.dtype == torch.int32
        b = ops.blocked_from_nchw(z)
        assert b.shape == (2, 2, 5, 7, 8) and b.dtype == torch.bfloat16
        assert ops.nchw_from_blocked(b, 8, True).dtype == torch.float32
        w = torch.empty(10, dtype=torch.int16)
        assert ops.conv2d_b16_fwd(b, 8, None, 0, w, c, c, c, None, None, geom, [0, 0, 0, 1, 0]).shape == (2, 2, 5, 7, 8)
        out = ops.conv2d_b16_fwd(b, 8, None, 0, w, c, c, c, None, None, geom, [0, 0, 0, 1, 1])
        assert out.shape == (2, 4, 5, 7) and out.dtype == torch.float32
        assert ops.blocked_from_nchw_sum(z, z).shape == (2, 2, 5, 7, 8)
        out = ops.conv2d_b16_nchw_fwd(b, 8, None, 0, w, c, c, c, None, None, torch.empty(2, 4, 5, 7), geom, [0, 0, 0, 1])
        assert out.shape == (2, 4, 5, 7) and out.dtype == torch.float32
        hw = torch.empty(2, 4)
        out = ops.conv2d_b16_head_fwd(b, 8, None, 0, w, c, c, c, None, None, hw, None, geom, [0, 0, 0, 1, 6, 1])
        assert out.shape == (2, 5, 7, 2) and out.dtype == torch.float32           # interleaved: a complex64 [2,5,7]
        assert ops.conv2d_b16_head_fwd(b, 8, None, 0, w, c, c, c, None, None, hw, None, geom, [0, 0, 0, 1, 6, 0]).shape == (2, 2, 5, 7)
        assert ops.adam_step(c, c, c, c, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1) is None
        assert ops.instnorm_act(z, None, None, 1e-5, 1, torch.empty(1), None).shape == z.shape
        m2, i2 = ops.bn_batch_stats(z, 1e-5)
        assert m2.shape == i2.shape == (8,)
        assert ops.bn_running_update(c, c.clone(), m2, i2, 70, 1e-5, 0.1) is None
        assert ops.gather_f32(c, torch.empty(12, dtype=torch.int32)).shape == (12,)
        # the differentiable block: a functional op whose backward is registered with the dispatcher
        w = torch.empty(4, 8, 3, 3, requires_grad=True)
        y, zz, mm, ii = ops.conv_norm_act(z, None, w, None, torch.empty(4), torch.empty(4), None, None, geom, False, 1,
                                          1e-5, 1, None)
        assert y.shape == zz.shape == (2, 4, 5, 7) and mm.shape == ii.shape == (4,)
        assert y.requires_grad and y.grad_fn is not None


def test_blocked_conv_folds_only_eval_batchnorm_and_handles_affine_false():
    import pytest
    import torch
    import torch.nn as nn
    from unet_amd.blocked import BlockedConv
    conv = nn.Conv2d(16, 32, (3, 3), padding=1)
    bn = nn.BatchNorm2d(32, affine=False)
    bn.running_mean.uniform_(-1, 1)
    bn.running_var.uniform_(0.5, 2)
    layer = BlockedConv(conv, bn, nn.ReLU())
    with pytest.raises(NotImplementedError):           # train mode: batch statistics cannot be folded
        layer._prepare(16, 0, torch.device("cpu"))
    bn.eval()
    st = layer._prepare(16, 0, torch.device("cpu"))
    scale = torch.rsqrt(bn.running_var + bn.eps)
    assert torch.equal(st["scale"], scale) and torch.equal(st["shift"], -bn.running_mean * scale)
    untracked = nn.BatchNorm2d(32, track_running_stats=False).eval()
    with pytest.raises(NotImplementedError):
        BlockedConv(conv, untracked, nn.ReLU())._prepare(16, 0, torch.device("cpu"))


def test_fp16_weight_image_matches_numpy_rounding():
    """The fp16 form of the weight image (UNETMI_B16_IN_F16): the same layout with nearest-even fp16 values -- normal,
    subnormal and tiny magnitudes as numpy rounds them, magnitudes beyond the largest finite fp16 value saturated --
    and the second half holding the fp16-rounded remainder."""
    import torch
    from unet_amd.blocked import pack_b16_weights
    rng = np.random.default_rng(11)
    K, C = 32, 16
    w = rng.standard_normal((K, C, 1, 1)).astype(np.float32)
    w[0, :, 0, 0] = np.float32(2.0) ** -np.arange(10, 26)                  # down into the subnormals and below them
    w[1, :, 0, 0] = (1.0 + 2.0 ** -11) * np.float32(2.0) ** np.arange(-8, 8)   # ties: to even
    w[2, :4, 0, 0] = [65504.0, 65519.9, 65520.0, -1e9]
    w[3, :3, 0, 0] = [6.0e-8, 2.9e-8, 3.1e-8]
    img = pack_b16_weights(torch.from_numpy(w), C, 0, [K, 1, 1, 1, 1, 0, 0, 1, 1, 0, 0], 0, halves=2, f16=True).numpy().view(np.uint16)
    img = img.reshape(1, 2, 1, 2, 32, 8)                                   # [group][weight half][tile][k half][row][k]
    with np.errstate(over="ignore"):
        hi = w.astype(np.float16)
    hi = np.where(np.isinf(hi), np.sign(w).astype(np.float16) * np.float16(65504), hi)
    with np.errstate(over="ignore"):
        lo = (w - hi.astype(np.float32)).astype(np.float16)
    for co in range(K):
        for ci in range(C):
            got = img[0, :, 0, ci // 8, co, ci % 8]
            assert got[0] == hi[co, ci, 0, 0].view(np.uint16), (co, ci, w[co, ci, 0, 0])
            if np.isfinite(lo[co, ci, 0, 0]) and abs(w[co, ci, 0, 0]) < 65520:
                assert got[1] == lo[co, ci, 0, 0].view(np.uint16), (co, ci)


def test_bf16_plan_takes_two_row_kernels_where_they_pay():
    """Host logic only (unetmi_conv2d_b16_kernel_name launches nothing).  DSUNet's benchmark shapes at batch 32: the
    (7,1) last decoder over 135 channels and the stride-(2,2) 7x5 decoder take the two-rows-per-wave kernels, the
    two-column-tile layers with two resident workgroups keep theirs, a layer with a fused head takes a one-column-tile
    kernel, and a small batch never takes the two-row kernels."""
    from unet_amd import _lib
    from unet_amd.ops import b16_desc, conv_b16_kernel_name

    def name(N, C0, C1, H, W, geom, head=0, f16=False):
        return conv_b16_kernel_name(b16_desc(N, C0, C1, H, W, geom, 1, _lib.ACT_LEAKY, 0, 0, 0, 1, False, f16, f16), head)

    dec9, dec8 = [90, 7, 1, 1, 1, 3, 0, 1, 1, 0, 0], [90, 1, 7, 1, 1, 0, 3, 1, 1, 0, 0]
    dec7 = [90, 7, 5, 2, 2, 3, 2, 1, 1, 0, 1]
    assert name(32, 90, 45, 257, 512, dec9) == "conv_b16_kernel_rt2<3, false>"
    assert name(32, 90, 45, 257, 512, dec9, head=2) == "conv_b16_kernel_occ2<3, 1, false>"
    assert name(32, 90, 45, 257, 512, dec8) == "conv_b16_kernel_occ2<3, 2, false>"
    assert name(32, 90, 90, 129, 256, dec7) == "conv_b16_kernel_pair_rt2<3, false>"
    assert name(2, 90, 90, 33, 128, dec7) == "conv_b16_kernel_pair_occ2<3, false>"
    # fp16 operands: the same plan, the instantiations on the fp16 matrix instruction
    assert name(32, 90, 45, 257, 512, dec9, head=2, f16=True) == "conv_b16_kernel_occ2<3, 1, true>"
    assert name(32, 90, 90, 129, 256, dec7, f16=True) == "conv_b16_kernel_pair_rt2<3, true>"
    with pytest.raises(_lib.UnetmiError, match="time-stride-2"):
        name(32, 90, 90, 129, 256, dec7, head=2)


def test_plan_picks_the_kernel_family_from_the_grid():
    """Host logic only (unetmi_conv2d_kernel_name launches nothing): which MFMA kernel family a layer gets depends on
    its grid, the packed weight image does not.  Shapes: the benchmark's 7x5 encoder, its 1x7 layer, its stride-2x2
    7x5 decoder and the stride-(2,1) decoder whose two row phases (65 and 64 rows) pad worse in ten-row tiles."""
    from unet_amd import _lib
    from unet_amd.ops import conv_kernel_name, make_desc

    def name(N, C0, C1, H, W, geom, transposed, rows=0):
        mode = _lib.IN_CONCAT if C1 else _lib.IN_SINGLE
        return conv_kernel_name(make_desc(N, C0, C1, H, W, geom, [transposed, mode, _lib.ACT_LEAKY, 0, rows]))

    def image_bytes(N, geom, rows=0):
        d = make_desc(N, 64, 0, 257, 509, geom, [0, _lib.IN_SINGLE, _lib.ACT_LEAKY, 0, rows])
        return _lib.lib.unetmi_conv2d_packed_weight_bytes(ctypes.byref(d))

    enc2 = [64, 7, 5, 2, 2, 3, 2, 1, 1, 0, 0]
    assert name(64, 64, 0, 257, 509, enc2, 0) == "conv_dma_kernel<1, 5, 2, 2, 5, 0>"      # 129 rows: 130 in tens
    assert name(64, 64, 0, 257, 509, enc2, 0, rows=4) == "conv_dma_kernel<1, 4, 2, 2, 5, 0>"
    assert name(64, 64, 0, 257, 509, enc2, 0, rows=6) == "conv_dma_kernel<1, 6, 2, 2, 5, 0>"
    assert name(1, 64, 0, 33, 64, enc2, 0) == "conv_dma_kernel<1, 1, 2, 2, 5, 0>"         # a small grid
    assert len({image_bytes(N, enc2, r) for N in (1, 8, 64) for r in (0, 1, 2, 4, 5, 6)}) == 1
    enc1 = [64, 1, 7, 1, 1, 0, 3, 1, 1, 0, 0]
    assert name(64, 32, 0, 257, 509, enc1, 0) == "conv_dma_kernel<1, 5, 2, 2, 7, 0>"      # 257 rows: 260 in tens
    dec7 = [64, 7, 5, 2, 2, 3, 2, 1, 1, 0, 0]
    assert name(64, 64, 64, 129, 255, dec7, 1) == "conv_dma_kernel<2, 5, 2, 2, 3, 2>"     # two column phases: five at most
    assert name(64, 64, 64, 129, 255, dec7, 1, rows=6) == "conv_dma_kernel<2, 4, 2, 2, 3, 2>"
    dec6 = [64, 7, 5, 2, 1, 3, 2, 1, 1, 0, 0]
    assert name(64, 64, 64, 65, 255, dec6, 1) == "conv_dma_kernel<1, 4, 2, 2, 5, 0>"      # 65 + 64 rows: 140 > 136
