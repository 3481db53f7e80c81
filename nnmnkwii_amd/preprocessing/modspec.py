"""Modulation spectrum of parameter trajectories on MI355X.

Host-side mirror of /root/reference/nnmnkwii/preprocessing/modspec.py (``modspec`` :6-53,
``modphase`` :57-58, ``inv_modspec`` :62-100, ``modspec_smoothing`` :103-167).  The reference
calls numpy's rfft / irfft along the time axis; here every feature column is one workgroup
of the LDS-resident FFT kernels behind ``include/mlpg_hip.h`` (``mlpg_hip_modspec*``), and a
``(B, T, D)`` batch is accepted wherever the reference takes ``(T, D)``.

Arithmetic is float64 on the device; results are cast to what numpy would return for the input
dtype (float32 in -> float32 / complex64 out).  DFT lengths must be powers of two up to 4096
(the reference's defaults and tests; anything else raises ``NotImplementedError`` -- there is no
CPU fallback).
"""
import numpy as np

from .. import _hip


def _norm_flag(norm):
    if norm is None or norm == "backward":
        return False
    if norm == "ortho":
        return True
    raise ValueError('Invalid norm value {}; should be None, "backward" or "ortho".'.format(norm))


def _check_n(n):
    n = int(n)
    if n < 2:
        raise ValueError("the DFT length must be at least 2, got %d" % n)
    return n


def _to_dev(x):
    """numpy (T, D) / (B, T, D) or CUDA tensor -> (float64 CUDA (B, T, D), was_numpy, had_batch, real dtype)."""
    torch = _hip.torch_mod()
    if torch.is_tensor(x):
        t = x
        rdt = np.float32 if x.dtype == torch.float32 else np.float64
        is_np = False
    else:
        x = np.asarray(x)
        rdt = np.float32 if x.dtype == np.float32 else np.float64
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(_hip.require_gpu())
        is_np = True
    batched = t.dim() == 3
    if not batched:
        assert t.dim() == 2
        t = t[None]
    return t, is_np, batched, rdt


def modspec(x, n=4096, norm=None, return_phase=False):
    """Modulation spectrum: power of the n-point DFT of the trajectory along time,
    ``(T, D) -> (n//2 + 1, D)`` (and the unit phasors ``exp(1j * angle)`` if ``return_phase``)."""
    torch = _hip.torch_mod()
    n = _check_n(n)
    t, is_np, batched, rdt = _to_dev(x)
    if t.shape[1] > n:
        t = t[:, :n]                                   # numpy's rfft(x, n) crops a longer input
    ms, ph = _hip.modspec(t, n, _norm_flag(norm), want_phase=return_phase)
    if not batched:
        ms = ms[0]
        ph = ph[0] if ph is not None else None
    if is_np:
        ms = ms.cpu().numpy().astype(rdt, copy=False)
        if return_phase:
            ph = ph.cpu().numpy()
            ph = (ph[..., 0] + 1j * ph[..., 1]).astype(np.complex64 if rdt == np.float32 else np.complex128, copy=False)
    elif return_phase:
        ph = torch.view_as_complex(ph)
    return (ms, ph) if return_phase else ms


def modphase(x, n=4096, norm=None):
    return modspec(x, n, norm, return_phase=True)[1]


def inv_modspec(ms, phase, norm=None):
    """Inverse of :func:`modspec`: ``irfft(sqrt(ms) * phase)``, ``(n//2 + 1, D) -> (n, D)``."""
    torch = _hip.torch_mod()
    is_np = not torch.is_tensor(ms)
    if is_np:
        ms_h = np.asarray(ms)
        rdt = np.float32 if ms_h.dtype == np.float32 else np.float64
        dev = _hip.require_gpu()
        ms_t = torch.from_numpy(np.ascontiguousarray(ms_h, dtype=np.float64)).to(dev)
        ph_h = np.asarray(phase).astype(np.complex128, copy=False)
        ph_t = torch.from_numpy(np.ascontiguousarray(np.stack([ph_h.real, ph_h.imag], axis=-1))).to(dev)
    else:
        rdt = np.float32 if ms.dtype == torch.float32 else np.float64
        ms_t = ms
        ph_t = torch.view_as_real(phase.to(torch.complex128).contiguous())
    batched = ms_t.dim() == 3
    if not batched:
        ms_t, ph_t = ms_t[None], ph_t[None]
    _check_n((ms_t.shape[1] - 1) * 2)
    out = _hip.inv_modspec(ms_t, ph_t, _norm_flag(norm))
    if not batched:
        out = out[0]
    return out.cpu().numpy().astype(rdt, copy=False) if is_np else out


def modspec_smoothing(x, modfs, n=4096, norm=None, cutoff=50, log_domain=True):
    """Smooth a trajectory by removing the modulation-frequency bands above ``cutoff`` Hz
    (forward DFT, band removal and inverse DFT fused in one launch; ``(T, D) -> (T, D)``)."""
    t, is_np, batched, rdt = _to_dev(x)
    T = t.shape[1]
    if cutoff is not None and cutoff > modfs // 2:          # modspec.py:145-150
        raise ValueError("Cutoff frequency {} hz must be larger than Nyquist freqeuency {}. hz".format(cutoff, modfs // 2))
    if n < T:                                               # modspec.py:151-154
        raise RuntimeError("DFT length {} must be larger than time length {}".format(n, T))
    n = _check_n(n)
    nb = n // 2 + 1
    limit_bin = nb if cutoff is None else min(int(n * cutoff / modfs) + 1, nb)   # :160-163
    if n % 2:
        # the reference inverts through inv_modspec, i.e. at length 2 (K - 1) = n - 1 for an odd n: same here,
        # composed from the three steps (the fused launch transforms back at n)
        ms, ph = _hip.modspec(t, n, _norm_flag(norm), want_phase=True)
        if limit_bin < nb:
            ms[:, limit_bin:] = 1.0 if log_domain else 0.0     # exp(0) / 0
        out = _hip.inv_modspec(ms, ph, _norm_flag(norm))[:, :T].contiguous()
    else:
        out = _hip.modspec_smoothing(t, n, limit_bin, log_domain, _norm_flag(norm))
    if not batched:
        out = out[0]
    return np.ascontiguousarray(out.cpu().numpy().astype(rdt, copy=False)) if is_np else out
