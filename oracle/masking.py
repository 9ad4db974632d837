"""Mask / ratio-filter application of the oracle.  Test infrastructure.

``resunet_mask_apply`` and ``crm_apply`` restate expressions that exist in the
reference.  ``crf_apply`` has no counterpart there: cRFUNet.py:315-324 only
EMITS the complex ratio filter and nothing in the reference applies it, so
its definition below is this project's own and its parity is UNPINNED (it is
checked against this restatement only).
"""

import torch


def resunet_mask_apply(y, x):
    """(tanh(y) + 1e-13) * x[:, :2]  -- ResUNet.py:217-219."""
    return torch.mul(torch.tanh(y) + 1e-13, x[:, :2, :, :])


def crm_apply(X, M):
    """Complex ratio mask: X * M  -- DSUNet.py:361-362."""
    return X * M


def crf_apply(H, X):
    """Complex ratio filter (parity unpinned, see module docstring).

    H: complex [B, N, C, 2L_f+1, 2L_t+1, F, T]  (the tensor UNet20.forward returns)
    X: complex [B, C, F, T]
    Y[b,n,c,f,t] = sum_{p,q} H[b,n,c,p,q,f,t] * X[b,c,f+p-L_f,t+q-L_t], zero outside.
    Returns complex [B, N, C, F, T].
    """
    B, N, C, PF, PT, F, T = H.shape
    L_f, L_t = (PF - 1) // 2, (PT - 1) // 2
    Xp = torch.nn.functional.pad(torch.view_as_real(X), (0, 0, L_t, L_t, L_f, L_f))
    Xp = torch.view_as_complex(Xp.contiguous())
    Y = torch.zeros((B, N, C, F, T), dtype=H.dtype)
    for p in range(PF):
        for q in range(PT):
            Y = Y + H[:, :, :, p, q] * Xp[:, None, :, p:p + F, q:q + T]
    return Y
