"""The entry point that allocates nothing (unetmi_bn_train_stats_static: partial sums in the library's own static
device array) against unetmi_bn_train_stats with a caller workspace -- same kernels, so the results must be the
same bits.  Two calls in a row, the second on a smaller input (the array keeps the first call's contents); then the
size limit."""

import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


_warmed = {}


def _both(lib, z, eps=1e-5, momentum=0.1):
    from unet_amd._lib import check
    N, C, HW = z.shape
    dev = z.device
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    for static in (False, True):
        rm = torch.linspace(-1.0, 1.0, C, device=dev)
        rv = torch.linspace(0.5, 1.5, C, device=dev)
        mean = torch.full((C,), float("nan"), device=dev)
        invstd = torch.full((C,), float("nan"), device=dev)
        if static:
            torch.cuda.synchronize()
            free_before, reserved_before = torch.cuda.mem_get_info()[0], torch.cuda.memory_reserved()
            check(lib.unetmi_bn_train_stats_static(N, C, HW, z.data_ptr(), eps, momentum, rm.data_ptr(), rv.data_ptr(),
                                                   mean.data_ptr(), invstd.data_ptr(), stream), "bn_train_stats_static")
            torch.cuda.synchronize()
            if _warmed.get(str(dev)):      # the first call on a device may load the code object; later ones allocate nothing
                assert torch.cuda.mem_get_info()[0] == free_before, "the static entry point took device memory"
                assert torch.cuda.memory_reserved() == reserved_before
            _warmed[str(dev)] = True
        else:
            nbytes = lib.unetmi_bn_workspace_bytes(N, C, HW)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(lib.unetmi_bn_train_stats(N, C, HW, z.data_ptr(), eps, momentum, rm.data_ptr(), rv.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), nbytes, stream),
                  "bn_train_stats")
        torch.cuda.synchronize()
        results.append((mean.cpu(), invstd.cpu(), rm.cpu(), rv.cpu()))
    return results


def test_static_scratch_entry_point_matches_workspace_one(device):
    from unet_amd import _lib
    lib = _lib.lib
    g = torch.Generator().manual_seed(7)
    # first call: a large input (many blocks per channel); second: a smaller one with fewer channels and blocks
    for shape in ((8, 64, 129 * 255), (3, 20, 1000), (1, 384, 501)):
        z = (torch.randn(shape, generator=g) * 2.0 + 0.5).to(device)
        (m0, i0, rm0, rv0), (m1, i1, rm1, rv1) = _both(lib, z)
        assert not torch.isnan(m1).any() and not torch.isnan(i1).any()
        assert torch.equal(m0, m1) and torch.equal(i0, i1), shape
        assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1), shape
        # and both are the statistics torch computes
        zc = z.cpu().double()
        want_mean = zc.mean(dim=(0, 2))
        want_var = zc.var(dim=(0, 2), unbiased=False)
        assert (m1.double() - want_mean).abs().max().item() <= 1e-5 * (want_mean.abs().max().item() + 1.0)
        assert ((i1.double() - 1.0 / torch.sqrt(want_var + 1e-5)).abs().max().item()
                <= 1e-5 * (1.0 / torch.sqrt(want_var + 1e-5)).max().item())


def test_static_scratch_entry_point_refuses_oversize(device):
    from unet_amd import _lib
    lib = _lib.lib
    C = 20000                                  # 20000 channels x >= 1 block x 16 bytes > 256 KiB
    z = torch.zeros((1, C, 4), device=device)
    out = torch.zeros((2, C), device=device)
    rc = lib.unetmi_bn_train_stats_static(1, C, 4, z.data_ptr(), 1e-5, 0.1, None, None, out[0].data_ptr(),
                                          out[1].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.E_UNSUPPORTED
    assert "static scratch" in _lib.last_error()
    assert ctypes.c_size_t(lib.unetmi_bn_workspace_bytes(1, C, 4)).value > 256 * 1024
