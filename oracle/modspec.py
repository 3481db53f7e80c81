"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's modulation-spectrum path (numpy's FFT is the
algorithm: the reference calls it directly).  Imported by tests/ only; never by the product path.

Follows /root/reference/nnmnkwii/preprocessing/modspec.py (modspec :6-53, inv_modspec :62-100, modspec_smoothing
:103-167) and the analytic gradient of /root/reference/nnmnkwii/autograd/_impl/modspec.py:30-60.  Pinned against the
goldens the reference itself produced (tests/golden/mlpg_golden.npz: modspec/*, modspec_grad/*) by
tests/test_oracle_cpu.py::test_modspec_oracle_matches_reference_goldens; the GPU tests then use it for DFT lengths the
goldens do not hold (non powers of two, n > 4096).
"""
import numpy as np


def modspec(x, n=4096, norm=None, return_phase=False):
    """Power of the one-sided DFT along time; phase as unit phasors (modspec.py:45-53)."""
    s = np.fft.rfft(np.asarray(x), n=n, axis=0, norm=norm)
    ms = s.real ** 2 + s.imag ** 2
    if return_phase:
        return ms, np.exp(1j * np.angle(s))
    return ms


def inv_modspec(ms, phase, norm=None):
    """irfft of sqrt(ms) * phase at n = 2 (K - 1) (modspec.py:88-100)."""
    n = (ms.shape[0] - 1) * 2
    return np.fft.irfft(np.sqrt(ms) * phase, n=n, axis=0, norm=norm)


def modspec_smoothing(x, modfs, n=4096, norm=None, cutoff=50, log_domain=True):
    """Zero the (log-)power of the bins from int(n cutoff / modfs) + 1 on and transform back (modspec.py:140-167).
    As in the reference the inverse goes through inv_modspec, i.e. at length 2 (K - 1): n for even n, n - 1 for odd n."""
    x = np.asarray(x)
    T = x.shape[0]
    if cutoff > modfs // 2:
        raise ValueError("cutoff above Nyquist")
    if n < T:
        raise RuntimeError("DFT length smaller than the time length")
    ms, ph = modspec(x, n=n, norm=norm, return_phase=True)
    if log_domain:
        with np.errstate(divide="ignore"):
            ms = np.log(ms)
    limit_bin = int(n * cutoff / modfs) + 1
    if limit_bin < len(ms):
        ms[limit_bin:] = 0
    if log_domain:
        ms = np.exp(ms)
    return np.ascontiguousarray(inv_modspec(ms, ph, norm=norm)[:T])


def modspec_grad(y, grad_ms, n, norm=None):
    """d sum(grad_ms * modspec(y)) / dy (autograd/_impl/modspec.py:36-60): C sum_k g_k (R_k cos + I_k sin)(-2 pi k t / n),
    C = 2 (2 / sqrt(n) for norm="ortho") -- on top of an ortho-scaled spectrum."""
    y = np.asarray(y, dtype=np.float64)
    T = min(y.shape[0], n)
    s = np.fft.rfft(y, n=n, axis=0, norm=norm)
    kt = -2.0 * np.pi / n * (np.arange(n // 2 + 1)[:, None] * np.arange(T)[None, :] % n)
    C = 2.0 / np.sqrt(n) if norm == "ortho" else 2.0
    gr = np.asarray(grad_ms, dtype=np.float64)
    out = np.zeros_like(y)
    out[:T] = C * (np.cos(kt).T @ (gr * s.real) + np.sin(kt).T @ (gr * s.imag))
    return out
