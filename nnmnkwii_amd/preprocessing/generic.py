"""Frame utilities on the DTW path (mirror of nnmnkwii/preprocessing/generic.py:291-332)."""
import numpy as np

from .. import _hip


def trim_zeros_frames(x, eps=1e-7, trim="b"):
    """Remove leading and/or trailing zero frames of a ``(T, D)`` feature matrix.

    Drop-in for ``nnmnkwii.preprocessing.trim_zeros_frames``: a frame is zero when
    ``sum_d |x| < eps``; ``trim`` is ``"b"`` (trailing, the default -- the only
    mode DTWAligner uses), ``"f"`` (leading) or ``"fb"``.  Returns a view of ``x``.
    The frame scan runs in the HIP kernel ``mlpg_hip_trim_lengths``.
    """
    assert trim in {"f", "b", "fb"}
    torch = _hip.torch_mod()
    x = np.asarray(x)
    T, D = x.shape
    dev = _hip.require_gpu()
    xf = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
    xt = torch.from_numpy(np.ascontiguousarray(xf)).to(dev)
    back = T
    front = 0
    if "b" in trim:
        back = int(_hip.trim_lengths(xt[None].contiguous(), eps)[0].item())
    if "f" in trim:
        front = T - int(_hip.trim_lengths(torch.flip(xt, dims=[0])[None].contiguous(), eps)[0].item())
    if trim == "b":
        return x if back == T else x[: back]
    if trim == "f":
        return x[front:]
    return x[front:back] if back > front else x[:0]


def _same_window(window):
    """np.correlate(x, window, "same") as a (l, u, coeff) triple: out[t] = sum_k coeff[l+k] x[t+k]."""
    w = np.asarray(window, dtype=np.float64).ravel()
    L = len(w)
    u = (L - 1) // 2
    return (L - 1 - u, u, w)


def delta_features(x, windows):
    """Compute delta features and combine them, ``(T, D) -> (T, D * len(windows))``.

    Drop-in for ``nnmnkwii.preprocessing.delta_features`` (preprocessing/generic.py:250-288):
    ``windows`` is a list of ``(l, u, coeff)`` triples (the paramgen convention; like the
    reference only ``coeff`` is used) or of plain coefficient arrays; each output block is the
    "same"-mode correlation of every feature dimension with the window.  A ``(B, T, D)`` batch is
    accepted as well (optionally as a CUDA tensor).  Runs in ``mlpg_hip_delta_features``.
    """
    torch = _hip.torch_mod()
    assert len(windows) > 0
    if isinstance(windows[0], tuple):
        wins = [_same_window(w[2]) for w in windows]
    else:
        wins = [_same_window(w) for w in windows]
    if torch.is_tensor(x):
        xt = x if x.dim() == 3 else x[None]
        out = _hip.delta_features(xt.contiguous(), wins)
        return out if x.dim() == 3 else out[0]
    x = np.asarray(x)
    dev = _hip.require_gpu()
    xf = x if x.dtype in (np.float32, np.float64) else x.astype(np.float64)
    if xf.shape[-2] < max(len(w[2]) for w in wins):
        raise ValueError("delta_features: windows longer than the sequence are not supported")
    xt = torch.from_numpy(np.ascontiguousarray(xf)).to(dev)
    out = _hip.delta_features(xt if xt.dim() == 3 else xt[None], wins)
    out = out.cpu().numpy().astype(x.dtype, copy=False)
    return out if x.ndim == 3 else out[0]
