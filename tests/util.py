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


def assert_close(a, b, tol=FP32_REL_TOL, what=""):
    assert tuple(a.shape) == tuple(b.shape), "{}: shape {} vs {}".format(what, tuple(a.shape), tuple(b.shape))
    e = rel_err(a, b)
    assert e <= tol, "{}: rel-err {:.3e} > {:.1e}".format(what, e, tol)
    return e


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def sample_index(numel, n=2048):
    """Same fixed indices as tests/golden/make_golden.py."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    step = numel / n
    return (np.arange(n) * step + (np.arange(n) * 7919) % max(int(step), 1)).astype(np.int64) % numel
