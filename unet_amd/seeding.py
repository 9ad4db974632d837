"""Deterministic, torch-independent fill of parameters, buffers and inputs.

A counter-based generator (splitmix64 over ``hash(name) + index``) so that any
two processes -- the reference in the build container, the CPU restatement on
the GPU box's host, the HIP modules on the device, every data-parallel rank --
regenerate bit-identical weights from a seed and a tensor name without
shipping checkpoints.  BatchNorm statistics and affines are randomised so that
folding them into a kernel epilogue is actually exercised.
"""

import math
import re

import numpy as np
import torch
import torch.nn as nn

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text, seed):
    h = (0xCBF29CE484222325 ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    for b in text.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform01(name, n, seed=0):
    """n float64 values in [0,1), a pure function of (name, seed, index)."""
    key = np.uint64(_fnv1a64(name, seed))
    with np.errstate(over="ignore"):
        ctr = key + np.arange(n, dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03)
    bits = _splitmix64(ctr) >> np.uint64(11)
    return bits.astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(name, shape, lo, hi, seed=0, dtype=torch.float32):
    n = int(np.prod(shape)) if len(shape) else 1
    v = lo + (hi - lo) * uniform01(name, n, seed)
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def normal(name, shape, std=1.0, seed=0, dtype=torch.float32):
    """Box-Muller on two uniform streams."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = np.maximum(uniform01(name + "/u1", n, seed), 1e-300)
    u2 = uniform01(name + "/u2", n, seed)
    v = std * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def seeded_fill_(module, seed=0):
    """Overwrite every parameter and buffer of ``module`` in place.

    Conv / ConvTranspose / Linear: U(-a, a) with a = sqrt(6 / fan) where fan is
    the number of products summed into one output element (for a transposed
    conv the taps that actually hit one output: C_in*kH*kW / (sH*sW)), which
    keeps activations O(1) through 20 layers.  Norm layers: weight U(0.5, 1.5),
    bias and running_mean U(-0.2, 0.2), running_var U(0.5, 1.5); the BatchNorm that
    closes a residual-block layer (``res_<i>.3``) gets a quarter of that weight range,
    because a block sums five such layers onto its running value and twelve blocks in a
    row would otherwise grow the activations by ~4x each (saturating ResUNet's tanh
    mask completely, which would leave a parity test nothing but signs to compare).  PReLU slope
    U(0.1, 0.4).  LSTM / GRU: U(-k, k), k = 1/sqrt(hidden).
    Shared tensors (one PReLU registered under several names) are written once.
    """
    done = set()
    # a tensor reachable under several state_dict keys (aliased modules) is
    # seeded by the smallest key, independent of registration order
    canon = {}
    for key, t in module.state_dict(keep_vars=True).items():
        canon[id(t)] = min(canon.get(id(t), key), key)

    def put(t, name, lo, hi):
        if id(t) in done:
            return
        done.add(id(t))
        name = canon.get(id(t), name)
        with torch.no_grad():
            t.copy_(uniform(name, tuple(t.shape), lo, hi, seed, t.dtype))

    for mname, m in module.named_modules(remove_duplicate=False):
        if isinstance(m, (nn.Conv2d, nn.Conv1d)):
            fan = m.weight[0].numel()
            a = math.sqrt(6.0 / fan)
        elif isinstance(m, nn.ConvTranspose2d):
            fan = m.in_channels * m.kernel_size[0] * m.kernel_size[1] / m.groups
            fan = max(fan / (m.stride[0] * m.stride[1]), 1.0)
            a = math.sqrt(6.0 / fan)
        elif isinstance(m, nn.Linear):
            a = math.sqrt(3.0 / m.in_features)
        else:
            a = None
        if a is not None:
            put(m.weight, mname + ".weight", -a, a)
            if m.bias is not None:
                put(m.bias, mname + ".bias", -0.1, 0.1)
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d, nn.InstanceNorm2d, nn.LayerNorm)):
            if getattr(m, "weight", None) is not None:
                g = 0.25 if re.search(r"(^|\.)res_\d+\.3$", mname) else 1.0
                put(m.weight, mname + ".weight", 0.5 * g, 1.5 * g)
                put(m.bias, mname + ".bias", -0.2, 0.2)
            if getattr(m, "running_mean", None) is not None:
                put(m.running_mean, mname + ".running_mean", -0.2, 0.2)
                put(m.running_var, mname + ".running_var", 0.5, 1.5)
        elif isinstance(m, nn.PReLU):
            put(m.weight, mname + ".weight", 0.1, 0.4)
        elif isinstance(m, (nn.LSTM, nn.GRU)):
            k = 1.0 / math.sqrt(m.hidden_size)
            for pname, p in m.named_parameters():
                put(p, mname + "." + pname, -k, k)
    return module


def stft_like_input(name, shape, seed=0):
    """Synthetic complex-STFT planes: N(0,1) re/im stacked on the channel axis."""
    return normal(name, shape, 1.0, seed)
