"""Padded-batch glue around the hot path.

Mirrors ``apply_each2d_trim`` / ``apply_each2d_padded`` of /root/reference/nnmnkwii/util/__init__.py:19-66:
apply a ``(T, D) -> (T', D')`` function to every utterance of a zero-padded ``(N, Tmax, D)`` array and
collect the results in a zero-padded ``(N, Tmax, D')`` float64 array.  When the function is one of this
package's batched operations (``paramgen.mlpg``, ``preprocessing.delta_features``) the whole batch goes
to the GPU in one call instead of N; any other callable is applied per utterance, as in the reference.
"""
import numpy as np

from . import _hip
from .paramgen import _mlpg as _pg
from .preprocessing import generic as _gen


def _batched(func2d, X, lengths, args, kwargs):
    """One-launch equivalents of the per-utterance loop, or None if func2d is not one of ours."""
    if kwargs:
        return None
    if func2d is _pg.mlpg and len(args) == 2:
        variances, windows = args
        variances = np.asarray(variances)
        if variances.ndim != 1:          # per-utterance (T, D) variances cannot be shared by the batch
            return None
        return _pg.mlpg_batch(X, variances, windows, lengths=lengths)
    if func2d is _gen.delta_features and len(args) == 1:
        torch = _hip.torch_mod()
        dev = _hip.require_gpu()
        Xf = np.ascontiguousarray(X if X.dtype in (np.float32, np.float64) else X.astype(np.float64))
        win = [_gen._same_window(w[2] if isinstance(w, tuple) else w) for w in args[0]]
        if len(lengths) and int(np.min(lengths)) < max(len(w[2]) for w in win):
            return None                  # np.correlate swaps its operands there; the per-utterance path raises
        out = _hip.delta_features(torch.from_numpy(Xf).to(dev), win,
                                  torch.as_tensor(np.asarray(lengths), dtype=torch.int32, device=dev))
        return out.cpu().numpy()
    return None


def apply_each2d_padded(func2d, X, lengths, *args, **kwargs):
    """Apply ``func2d`` to ``X[n][:lengths[n]]`` for every n (util/__init__.py:44-66)."""
    assert X.ndim == 3
    N, T, _ = X.shape
    lengths = np.asarray(lengths)
    fast = _batched(func2d, X, lengths, args, kwargs)
    if fast is not None:
        return fast.astype(np.float64, copy=False)      # the reference collects into np.zeros (float64)
    y = func2d(X[0][: lengths[0]], *args, **kwargs)
    assert y.ndim == 2
    Y = np.zeros((N, T, y.shape[1]))
    Y[0][: len(y)] = y
    for idx in range(1, N):
        y = func2d(X[idx][: lengths[idx]], *args, **kwargs)
        Y[idx][: len(y)] = y
    return Y


def apply_each2d_trim(func2d, X, *args, **kwargs):
    """Apply ``func2d`` to every utterance with its trailing zero frames removed (util/__init__.py:19-41).
    The trim of the whole batch is one ``mlpg_hip_trim_lengths`` launch."""
    assert X.ndim == 3
    torch = _hip.torch_mod()
    dev = _hip.require_gpu()
    Xf = np.ascontiguousarray(X if X.dtype in (np.float32, np.float64) else X.astype(np.float64))
    lengths = _hip.trim_lengths(torch.from_numpy(Xf).to(dev)).cpu().numpy()
    return apply_each2d_padded(func2d, X, lengths, *args, **kwargs)
