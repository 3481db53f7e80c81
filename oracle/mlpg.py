"""MLPG oracle (CPU) -- TEST INFRASTRUCTURE ONLY, never imported by the product.

``mlpg`` / ``mlpg_batch`` call the C restatement in ``mlpg_oracle.c`` (built by
``oracle/Makefile`` into ``oracle/liboracle.so``).  ``mlpg_grad``,
``unit_variance_mlpg_matrix`` and ``reshape_means`` are numpy restatements for
small sizes.  All follow /root/reference/nnmnkwii/paramgen/_mlpg.py; line
numbers are cited per function.

Parity status: PINNED (tests/test_oracle.py vs tests/golden/*.npz produced by
the reference itself, plus SURVEY.md 8(c)'s known-answer vector).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mlpg_oracle.c", "dtw_oracle.c")]
    if (not force and os.path.exists(so)
            and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs)):
        return so
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        vp, i32p, f64p = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
        lng, ci = ctypes.c_long, ctypes.c_int
        for name in ("oracle_mlpg_f64", "oracle_mlpg_f32"):
            f = getattr(L, name)
            f.restype = lng
            f.argtypes = [vp, vp, ci, i32p, lng, lng, lng, lng, i32p, i32p, f64p, vp, i32p]
        L.oracle_mlpg_stages_f64.restype = lng
        L.oracle_mlpg_stages_f64.argtypes = [vp, vp, lng, lng, lng, lng, i32p, i32p, f64p, vp, vp, vp]
        _LIB = L
    return _LIB


def pack_windows(windows):
    """(l, u, coeff) triples -> int32 l[], int32 u[], packed float64 coeff[]."""
    wl = np.ascontiguousarray([int(w[0]) for w in windows], dtype=np.int32)
    wu = np.ascontiguousarray([int(w[1]) for w in windows], dtype=np.int32)
    for l, u, c in windows:
        assert l >= 0 and u >= 0                      # _mlpg.py:44
        assert len(c) == l + u + 1                    # _mlpg.py:45
    wc = np.ascontiguousarray(np.concatenate([np.asarray(w[2], dtype=np.float64).ravel() for w in windows]))
    return wl, wu, wc


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleLinAlgError(np.linalg.LinAlgError):
    pass


def mlpg_batch(means, variances, windows, lengths=None):
    """Loop of paramgen.mlpg over a zero-padded (B, Tmax, D) batch.

    Mirrors how the reference batches (util/__init__.py:44-66).  ``variances``
    is (B, Tmax, D) or a global (D,).  Returns ((B, Tmax, sd) array, status).
    """
    means = np.ascontiguousarray(means)
    dtype = means.dtype
    if dtype not in (np.float32, np.float64):
        dtype = np.dtype(np.float64)
        means = means.astype(dtype)
    variances = np.ascontiguousarray(variances, dtype=dtype)
    B, Tmax, D = means.shape
    nw = len(windows)
    sd = D // nw
    var_is_global = int(variances.ndim == 1)
    if var_is_global:
        assert variances.shape == (D,)
    else:
        assert variances.shape == means.shape          # _mlpg.py:171
    wl, wu, wc = pack_windows(windows)
    out = np.zeros((B, Tmax, sd), dtype=dtype)
    status = np.zeros((B, sd), dtype=np.int32)
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        assert lengths.shape == (B,)
    fn = lib().oracle_mlpg_f64 if dtype == np.float64 else lib().oracle_mlpg_f32
    rc = fn(_ptr(means), _ptr(variances), var_is_global, _ptr(lengths), B, Tmax, D, nw,
            _ptr(wl), _ptr(wu), _ptr(wc), _ptr(out), _ptr(status))
    return out, status, rc


def mlpg(mean_frames, variance_frames, windows):
    """paramgen.mlpg restated (_mlpg.py:92-199). (T, D) -> (T, sd), input dtype."""
    mean_frames = np.asarray(mean_frames)
    variance_frames = np.asarray(variance_frames)
    T, D = mean_frames.shape
    if not (variance_frames.ndim == 1 and variance_frames.shape[0] == D):
        assert mean_frames.shape == variance_frames.shape
        variance_frames = variance_frames[None]
    out, status, rc = mlpg_batch(mean_frames[None], variance_frames, windows)
    if rc:
        k = int(status.ravel()[rc - 1])
        raise OracleLinAlgError("%d-th leading minor not positive definite" % k)
    return out[0].astype(mean_frames.dtype, copy=False)


def mlpg_stages(mean_frames, variance_frames, windows, d=0):
    """b, precision band (2*sdw+1, T) and lower Cholesky rectangle of static dim d."""
    mean_frames = np.ascontiguousarray(mean_frames, dtype=np.float64)
    variance_frames = np.ascontiguousarray(variance_frames, dtype=np.float64)
    T, D = mean_frames.shape
    if variance_frames.ndim == 1:
        variance_frames = np.ascontiguousarray(np.tile(variance_frames, (T, 1)))
    wl, wu, wc = pack_windows(windows)
    sdw = int(max(l + u for l, u, _ in windows))
    b = np.zeros(T)
    prec = np.zeros((2 * sdw + 1, T))
    chol = np.zeros((sdw + 1, T))
    bad = lib().oracle_mlpg_stages_f64(_ptr(mean_frames), _ptr(variance_frames), T, D, d, len(windows),
                                       _ptr(wl), _ptr(wu), _ptr(wc), _ptr(b), _ptr(prec), _ptr(chol))
    return b, prec, chol, bad


# ---------------------------------------------------------------- numpy parts

def window_matrix(l, u, coeff, T):
    """Dense T x T window matrix: W[t, t+k] = coeff[l+k] (build_win_mats, _mlpg.py:13-50)."""
    W = np.zeros((T, T))
    for k in range(-l, u + 1):
        for t in range(max(0, -k), min(T, T - k)):
            W[t, t + k] = coeff[l + k]
    return W


def _edge_mask(T, mw):
    m = np.ones(T)
    m[:mw] = 0                                         # _mlpg.py:192 / :354
    m[-mw:] = 0                                        # python "-0:" == whole axis
    return m


def mlpg_grad(mean_frames, variance_frames, windows, grad_output):
    """paramgen.mlpg_grad restated (_mlpg.py:202-281), dense small-T form.

    grads[:, w*sd+d] = o_d^T (P_d^-1 W_w^T diag(tau_{w,d})), float32 (T, D).
    The reference solves with LAPACK dgbsv on a dense T x T right-hand side
    (:275); here the same linear system is solved densely with numpy.
    """
    mean_frames = np.asarray(mean_frames)
    variance_frames = np.asarray(variance_frames)
    grad_output = np.asarray(grad_output)
    T, D = mean_frames.shape
    nw = len(windows)
    sd = D // nw
    mw = int(max(max(l, u) for l, u, _ in windows))
    Ws = [window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for l, u, c in windows]
    grads = np.zeros((T, D), dtype=np.float32)
    for d in range(sd):
        prec = np.zeros((nw, T))
        for w in range(nw):
            prec[w] = 1 / variance_frames[:, w * sd + d]       # input-dtype reciprocal, :259
            if w != 0:
                prec[w, :mw] = 0
                prec[w, -mw:] = 0
        P = sum(Ws[w].T @ np.diag(prec[w]) @ Ws[w] for w in range(nw))
        for w in range(nw):
            r = Ws[w].T @ np.diag(prec[w])
            grad = np.linalg.solve(P, r)
            grads[:, w * sd + d] = grad_output[:, d].T.dot(grad)
    return grads


def unit_variance_mlpg_matrix(windows, T):
    """paramgen.unit_variance_mlpg_matrix restated (_mlpg.py:297-373): float32 (T, nw*T)."""
    mw = int(max(max(l, u) for l, u, _ in windows))
    mask = _edge_mask(T, mw)
    Ws = [window_matrix(l, u, np.asarray(c, dtype=np.float64), T) for l, u, c in windows]
    mod = [Ws[0]] + [np.diag(mask) @ W for W in Ws[1:]]    # :357-367
    P = sum(m.T @ W for m, W in zip(mod, Ws))
    Pinv = np.linalg.inv(P)                                # :369-370 (banded inverse)
    return Pinv.dot(np.concatenate(mod, axis=0).T).astype(np.float32)   # :372-373


def reshape_means(means, static_dim):
    """paramgen.reshape_means restated (_mlpg.py:376-405)."""
    T, D = means.shape
    if D == static_dim:
        return means
    return means.reshape(T, -1, static_dim).transpose(1, 0, 2).reshape(-1, static_dim)


def delta_features(x, windows):
    """preprocessing.delta_features restated (preprocessing/generic.py:229-288): per feature
    dimension, np.correlate(x[:, d], window, "same"); tuple windows use their coefficient array."""
    x = np.asarray(x)
    T, D = x.shape
    out = np.empty((T, D * len(windows)), dtype=x.dtype)
    for idx, w in enumerate(windows):
        win = w[2] if isinstance(w, tuple) else w
        for d in range(D):
            out[:, D * idx + d] = np.correlate(x[:, d], win, mode="same")
    return out
