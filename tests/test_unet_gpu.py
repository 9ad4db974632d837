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
