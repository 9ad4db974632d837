"""Shared helpers of the test-suite."""

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: outputs match the reference CPU forward within
# rel-err 1e-4 in fp32.  rel-err = max|a-b| / max|b| over the tensor.
FP32_REL_TOL = 1e-4


def rel_err(a, b):
    a = torch.as_tensor(a).detach().cpu()
    b = torch.as_tensor(b).detach().cpu()
    if a.is_complex():
        a = torch.view_as_real(a)
    if b.is_complex():
        b = torch.view_as_real(b)
    a, b = a.double(), b.double()
    denom = max(b.abs().max().item(), 1e-30)
    return (a - b).abs().max().item() / denom


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 over the tensor: unlike the max-norm figure it is not dominated by the largest
    output, so an error spread over the many small values (masks near 0) shows up in it."""
    a = torch.as_tensor(a).detach().cpu()
    b = torch.as_tensor(b).detach().cpu()
    if a.is_complex():
        a = torch.view_as_real(a)
    if b.is_complex():
        b = torch.view_as_real(b)
    a, b = a.double(), b.double()
    return (a - b).norm().item() / max(b.norm().item(), 1e-30)


def assert_close(a, b, tol=FP32_REL_TOL, what=""):
    """Both error figures are held to the bar: max-norm (north_star's rel-err) and relative L2."""
    assert tuple(a.shape) == tuple(b.shape), "{}: shape {} vs {}".format(what, tuple(a.shape), tuple(b.shape))
    e = rel_err(a, b)
    e2 = rel_l2(a, b)
    print("{}: rel-err max-norm {:.2e}, L2 {:.2e} (bar {:.0e})".format(what or "tensor", e, e2, tol))
    assert e <= tol, "{}: rel-err {:.3e} > {:.1e}".format(what, e, tol)
    assert e2 <= tol, "{}: relative L2 error {:.3e} > {:.1e}".format(what, e2, tol)
    return e


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def sample_index(numel, n=2048):
    """Same fixed indices as tests/golden/make_golden.py."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    step = numel / n
    return (np.arange(n) * step + (np.arange(n) * 7919) % max(int(step), 1)).astype(np.int64) % numel
