"""Objective metrics the alignment path takes as ``dist`` callables.

Host-side mirror of /root/reference/nnmnkwii/metrics/__init__.py:27-71 (``melcd``) for numpy inputs: not a hot
path by itself, but ``DTWAligner(dist=melcd)`` is how the reference's own test suite customises the alignment
(tests/test_preprocessing.py:496-501), and the aligner recognises this function and evaluates the same local cost
on the GPU (include/mlpg_hip.h MLPG_HIP_DIST_SCALED_L2_NP).
"""
import math

import numpy as np

_logdb_const = 10.0 / np.log(10.0) * np.sqrt(2.0)   # metrics/__init__.py:5


def melcd(X, Y, lengths=None):
    """Mel-cepstrum distortion in dB between time-aligned sequences of shape (D,), (T, D) or (B, T, D)
    (with ``lengths`` for padded mini-batches).  Same arithmetic as the reference for numpy arrays."""
    if lengths is None:
        z = X - Y
        r = (z * z).sum(-1)
        r = math.sqrt(r) if np.isscalar(r) else np.sqrt(r)
        if not np.isscalar(r):
            r = r.mean()
        return _logdb_const * float(r)
    if len(X.shape) == 2:
        X, Y = X[:, :, None], Y[:, :, None]
    s = 0.0
    T = float(np.sum(lengths))
    for x, y, length in zip(X, Y, lengths):
        x, y = x[:length], y[:length]
        z = x - y
        s += np.sqrt((z * z).sum(-1)).sum()
    return _logdb_const * float(s) / T
