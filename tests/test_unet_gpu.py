"""End-to-end parity of the device UNet against the oracle (itself pinned
bit-exact to the reference) and against the committed golden vectors."""

import numpy as np
import pytest
import torch

from tests.util import assert_close, load_golden

pytestmark = pytest.mark.gpu


def _pair(f_dim, c_in, c_last, seed):
    import oracle
    import unet_amd
    ref = oracle.UNet(*oracle.default_unet(c_in=c_in, c_last=c_last, f_dim=f_dim)).eval()
    unet_amd.seeded_fill_(ref, seed=seed)
    dev = unet_amd.UNet(*unet_amd.default_unet_blocks(c_in=c_in, c_last=c_last, f_dim=f_dim))
    dev.load_state_dict(ref.state_dict(), strict=True)
    return ref, dev.eval()


def test_narrow_unet_matches_golden_and_oracle(device):
    from unet_amd.seeding import normal
    ref, dev = _pair(8, 2, 1, seed=11)
    x = normal("unet_narrow/x", (1, 2, 257, 29), seed=11)
    with torch.no_grad():
        want = ref(x)
        got = dev.to(device)(x.to(device))
    golden = torch.from_numpy(load_golden("unet_narrow_eval")["out"])
    assert_close(want, golden, tol=1e-5, what="oracle vs golden")
    assert_close(got, golden, what="HIP vs golden")
    assert_close(got, want, what="HIP vs oracle")


@pytest.mark.parametrize("T", [13, 45])
def test_full_width_unet_small_T(device, T):
    """The real channel widths (f_dim=32) at the smallest legal T (T = 13 mod 16)."""
    from unet_amd.seeding import normal
    ref, dev = _pair(32, 2, 1, seed=5)
    x = normal("unet_full/x", (2, 2, 257, T), seed=5)
    with torch.no_grad():
        want = ref(x)
        got = dev.to(device)(x.to(device))
    assert_close(got, want, what="UNet f_dim=32 T={}".format(T))


def test_batch_independence(device):
    """Eval-mode forward has no cross-sample op: a batch equals its samples run
    one at a time (the property that lets the batch shard over GPUs)."""
    from unet_amd.seeding import normal
    _, dev = _pair(8, 2, 1, seed=3)
    dev.to(device)
    x = normal("unet_batch/x", (3, 2, 257, 29), seed=3).to(device)
    with torch.no_grad():
        whole = dev(x)
        parts = torch.cat([dev(x[i:i + 1]) for i in range(3)])
    assert torch.equal(whole, parts)


LAYER_NAMES = ["encoder_{}".format(i) for i in range(10)] + ["decoder_{}".format(i) for i in range(10)]


def test_benchmark_size_batch_against_golden_and_oracle(device):
    """The benchmark workload itself: batch 64 x [2,257,509], full channel widths.
    Sample 0 is the golden input, so every layer's output for it is checked against the
    reference's per-layer checksums and samples; two other samples are checked against the
    oracle run on them alone (the batch is independent per sample)."""
    from unet_amd.seeding import normal
    from tests.util import sample_index
    ref, dev = _pair(32, 2, 1, seed=13)
    dev.to(device)
    x = normal("unet_full509/batch", (64, 2, 257, 509), seed=14)
    x[0] = normal("unet_full509/x", (1, 2, 257, 509), seed=13)[0]
    store = {}
    mods = dict(dev.named_modules())
    handles = [mods[n].register_forward_hook(lambda m, i, o, n=n: store.__setitem__(n, o[0].detach().cpu()))
               for n in LAYER_NAMES]
    with torch.no_grad():
        got = dev(x.to(device)).cpu()
    for h in handles:
        h.remove()
    golden = load_golden("unet_full_eval_T509")
    for n in LAYER_NAMES + ["out"]:
        t = (got[0] if n == "out" else store[n]).reshape(-1).double().numpy()
        assert t.size == int(golden[n + "_numel"]), n
        want = golden[n + "_samples"].astype(np.float64)
        have = t[sample_index(t.size)]
        scale = np.sqrt(float(golden[n + "_sumsq"]) / t.size)          # rms of the layer output
        assert np.abs(have - want).max() <= 1e-4 * max(np.abs(want).max(), scale), n
        assert abs((t * t).sum() - float(golden[n + "_sumsq"])) <= 1e-4 * float(golden[n + "_sumsq"]), n
        assert abs(t.sum() - float(golden[n + "_sum"])) <= 1e-4 * scale * t.size, n
    with torch.no_grad():
        for i in (31, 63):
            assert_close(got[i:i + 1], ref(x[i:i + 1]), what="sample {} of the batch".format(i))


def _bf16_errors(got, want):
    if got.is_complex():
        got, want = torch.view_as_real(got), torch.view_as_real(want)
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    return ((got - want).norm() / want.norm()).item(), ((got - want).abs().max() / want.abs().max()).item()


def test_unet_and_unet20_bf16_operands(device):
    """A bfloat16 input through the whole concat shell with 16-bit operands (channel-blocked layout, the default for
    a bfloat16 input), against the fp32 oracle on the same (bf16-representable) input.  With fp16 operands behind the
    first encoder (the default) both error figures are held to the bfloat16 bar of 1e-2; with bfloat16 operands
    throughout, or bfloat16 storage on the NCHW kernels, only relative L2 is (their max-norm is 2e-2..4e-2 on UNet)."""
    import oracle
    import unet_amd
    from unet_amd.seeding import normal, uniform
    ref, dev = _pair(32, 2, 1, seed=5)
    dev.to(device)
    x = normal("unet_bf16/x", (2, 2, 257, 45), seed=6).to(torch.bfloat16)
    with torch.no_grad():
        want = ref(x.float())
        assert dev.bf16_matrix is True and dev.operand_format == "fp16"
        dev.bf16_matrix = False
        stor = dev(x.to(device))
        dev.bf16_matrix = True
        got = dev(x.to(device))
        dev.operand_format = "bf16"
        got_b = dev(x.to(device))
    assert got.dtype == torch.bfloat16 and dev._blocked[1], "the 16-bit-operand path was not taken"
    l2, mx = _bf16_errors(got, want)
    l2b, mxb = _bf16_errors(got_b, want)
    l2s, mxs = _bf16_errors(stor, want)
    print("UNet fp16 operands behind enc0: relative L2 {:.3e}, max-norm {:.3e}".format(l2, mx))
    print("UNet bf16 operands: relative L2 {:.3e}, max-norm {:.3e}".format(l2b, mxb))
    print("UNet bf16 storage : relative L2 {:.3e}, max-norm {:.3e}".format(l2s, mxs))
    assert l2 <= 1e-2 and mx <= 1e-2 and l2b <= 1e-2 and l2s <= 1e-2

    ref20 = oracle.UNet20().eval()
    unet_amd.seeded_fill_(ref20, seed=21)
    dev20 = unet_amd.UNet20()
    dev20.load_state_dict(ref20.state_dict(), strict=True)
    ref20.T = dev20.T = 29
    dev20.eval().to(device)
    x20 = uniform("unet20_bf16/x", (2, 14, 257, 29), 0.0, 1.0, seed=22).to(torch.bfloat16)
    with torch.no_grad():
        want20 = ref20(x20.float())
        dev20.bf16_matrix = False
        stor20 = dev20(x20.to(device))
        dev20.bf16_matrix = True
        got20 = dev20(x20.to(device))
        dev20.operand_format = "bf16"
        got20_b = dev20(x20.to(device))
    assert got20.dtype == torch.complex64 and tuple(got20.shape) == tuple(want20.shape) and dev20._blocked[1]
    l2, mx = _bf16_errors(got20, want20)
    l2b, mxb = _bf16_errors(got20_b, want20)
    l2s, mxs = _bf16_errors(stor20, want20)
    print("UNet20 fp16 operands behind enc0: relative L2 {:.3e}, max-norm {:.3e}".format(l2, mx))
    print("UNet20 bf16 operands: relative L2 {:.3e}, max-norm {:.3e}".format(l2b, mxb))
    print("UNet20 bf16 storage : relative L2 {:.3e}, max-norm {:.3e}".format(l2s, mxs))
    assert l2 <= 2e-3 and mx <= 2e-3 and l2b <= 1e-2 and l2s <= 1e-2
