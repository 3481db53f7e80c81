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
