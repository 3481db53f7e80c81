"""MLPG on MI355X behind the signatures of ``nnmnkwii.paramgen``.

Host-side mirror of /root/reference/nnmnkwii/paramgen/_mlpg.py.  All numerical
work is done by the HIP kernels in ``nnmnkwii_amd/csrc`` through the C ABI in
``include/mlpg_hip.h``; this module only validates arguments the way the
reference does, moves arrays to the GPU and maps kernel status words to the
reference's exceptions.
"""
import numpy as np

from .. import _hip

__all__ = [
    "build_win_mats",
    "mlpg",
    "mlpg_grad",
    "unit_variance_mlpg_matrix",
    "reshape_means",
    "full_window_mat",
    "mlpg_batch",
    "multi_stream_mlpg",
]


class BandMat(object):
    """Minimal stand-in for the reference's bandmat.BandMat container
    (paramgen/_bandmat/core.pyx:20-87): ``l``, ``u``, ``data``, ``transposed``,
    ``.T``, ``.size`` and ``.full()``.  LAPACK band layout:
    ``full[j + i, j] == data[u + i, j]`` for ``i in [-u, l]``.
    """

    def __init__(self, l, u, data, transposed=False):
        self.l = l
        self.u = u
        self.data = data
        self.transposed = transposed
        assert self.l >= 0 and self.u >= 0
        assert self.data.ndim == 2 and self.data.shape[0] == self.l + self.u + 1

    def __repr__(self):
        return "BandMat(%r, %r, %r, transposed=%r)" % (self.l, self.u, self.data, self.transposed)

    @property
    def size(self):
        return self.data.shape[1]

    @property
    def T(self):
        return BandMat(self.u, self.l, self.data, transposed=not self.transposed)

    def full(self):
        l, u = (self.u, self.l) if self.transposed else (self.l, self.u)
        n = self.size
        m = np.zeros((n, n))
        for i in range(-u, l + 1):
            j = np.arange(max(0, -i), max(0, n + min(0, -i)))
            m[j + i, j] = self.data[u + i, j]
        return m.T if self.transposed else m


def build_win_mats(windows, T):
    """Window matrices as banded containers (reference: paramgen/_mlpg.py:13-50).

    ``W[t, t+k] = win_coeff[l+k]`` for ``k in [-l, u]``, one ``T x T`` Toeplitz
    band per window.
    """
    win_mats = []
    for l, u, win_coeff in windows:
        assert l >= 0 and u >= 0
        assert len(win_coeff) == l + u + 1
        data = np.tile(np.reshape(np.asarray(win_coeff, dtype=np.float64), (l + u + 1, 1)), T)
        win_mats.append(BandMat(u, l, data).T)
    return win_mats


def full_window_mat(win_mats, T):
    """Concatenated dense window matrix ``(T*num_windows, T)`` (paramgen/_mlpg.py:284-294)."""
    return np.concatenate([w.full() for w in win_mats], axis=0) if len(win_mats) else np.zeros((0, T))


def _as_float(a):
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return a


def _match_dtypes(m, v):
    """Means and variances of different float dtypes, as the reference treats them (_mlpg.py:180-188): the reciprocal
    is taken in the VARIANCES' dtype, everything after it in float64, the result is cast to the means' dtype.
    float64 means + float32 variances: the float32 reciprocal is taken here (one elementwise pass) and handed to the
    float64 kernel as an exactly invertible float64 variance; float32 means + float64 variances: float64 kernel."""
    torch = _hip.torch_mod()
    if v.dtype == m.dtype:
        return m, v
    if m.dtype == torch.float64 and v.dtype == torch.float32:
        tau32 = torch.reciprocal(v)                        # float32 reciprocal, correctly rounded
        return m, torch.reciprocal(tau32.to(torch.float64))
    return m.to(torch.float64), v.to(torch.float64)


def mlpg_batch(means, variances, windows, lengths=None, algo=_hip.ALGO_AUTO, check=True, device=None):
    """Batched MLPG over a zero-padded ``(B, Tmax, D)`` batch -- the GPU-native
    form of the reference's per-utterance loop (util/__init__.py:44-66).

    ``means``/``variances`` are numpy arrays or CUDA tensors (float32/float64);
    ``variances`` may be ``(B, Tmax, D)``, a global ``(D,)`` or ``None`` (unit
    variances).  ``lengths`` (B,) gives the valid frames per utterance; output
    frames beyond it are zero.  Returns the same kind of object as ``means``,
    shape ``(B, Tmax, D // len(windows))``.  numpy input: ``device`` may also be
    a list of GPU indices or ``"all"`` -- the utterance chunks are then dealt
    round-robin over those devices from this one process (mlpg_hip_forward_host_multi).
    """
    if isinstance(means, np.ndarray) or not hasattr(means, "is_cuda"):
        return _mlpg_batch_host(means, variances, windows, lengths, algo, check, device)
    torch = _hip.torch_mod()
    dev = _hip.require_gpu(means.device if means.is_cuda else device)
    m = means.to(dev).contiguous()
    assert m.dim() == 3
    out_dtype = m.dtype
    if variances is None:
        v = None
    else:
        v = (variances if torch.is_tensor(variances) else
             torch.from_numpy(np.ascontiguousarray(_as_float(variances)))).to(dev).contiguous()
        m, v = _match_dtypes(m, v)
    if v is not None and v.dim() != 1:
        assert v.shape == m.shape                         # paramgen/_mlpg.py:171
    L = None
    if lengths is not None:
        L = torch.as_tensor(np.asarray(lengths) if not torch.is_tensor(lengths) else lengths).to(
            device=dev, dtype=torch.int32).contiguous()
    out, status = _hip.forward(m, v, windows, L, algo=algo, want_status=check)
    if check:
        _hip.raise_on_status(status, out.shape[-1])
    if out.dtype != out_dtype:
        out = out.to(out_dtype)                            # output dtype = dtype of the means (_mlpg.py:166,183)
    return out


def _mlpg_batch_host(means, variances, windows, lengths, algo, check, device):
    """numpy in -> numpy out through the host-pointer entry point of the C ABI (no torch): chunked, with the PCIe
    transfers overlapped with the kernels (include/mlpg_hip.h mlpg_hip_forward_host)."""
    m = np.ascontiguousarray(_as_float(means))
    assert m.ndim == 3
    out_dtype = m.dtype
    v = None
    if variances is not None:
        v = np.ascontiguousarray(_as_float(variances))
        if v.ndim != 1:
            assert v.shape == m.shape                      # paramgen/_mlpg.py:171
        if v.dtype != m.dtype:                             # see _match_dtypes: reciprocal in the variances' dtype
            if m.dtype == np.float64:
                v = 1.0 / (np.float32(1.0) / v).astype(np.float64)
            else:
                m = m.astype(np.float64)
    # default: the process's current GPU, not GPU 0; a list of indices or "all": the chunks dealt over those devices
    out, status = _hip.forward_host(m, v, windows, lengths, algo=algo, device=device)
    if check and status.any():
        _raise_host_status(status)
    return out if out.dtype == out_dtype else out.astype(out_dtype)


def _raise_host_status(status):
    """The reference's LinAlgError (linalg.pyx:79-82) for the first failing (utterance, dim) system of a host-memory call."""
    st = status.ravel()
    bad = np.flatnonzero(st)
    if bad.size:
        k = int(st[bad[0]])
        if k > 0:
            raise np.linalg.LinAlgError("%d-th leading minor not positive definite" % k)
        raise np.linalg.LinAlgError("the blocked elimination broke down (numerically singular system)" if k == -2
                                    else "internal error: inter-workgroup wait timed out")


def multi_stream_mlpg(inputs, variances, windows, stream_sizes, has_dynamic_features, lengths=None,
                      algo=_hip.ALGO_AUTO, check=True, device=None):
    """MLPG over every stream of a multi-stream acoustic feature matrix in ONE call.

    The Merlin-style layout the reference's data sources use (util/files.py:90-115): the feature
    axis is ``[stream 0 | stream 1 | ...]`` with ``stream_sizes`` columns each (e.g. mgc 180 | lf0 3 |
    vuv 1 | bap 15); a stream with dynamic features is window-major (static, delta, delta-delta) and
    is replaced by its maximum-likelihood static trajectory ``paramgen.mlpg(stream, var, windows)``;
    a stream without (``has_dynamic_features[k]`` False, e.g. vuv) is copied through.  This is the
    per-stream, per-utterance loop users write with ``util.apply_each2d_padded`` (util/__init__.py:
    44-66), done on the GPU directly on the padded ``(N, Tmax, D)`` batch, each stream consumed in
    place (no slicing copies).

    inputs: ``(T, D)`` or ``(N, Tmax, D)`` numpy array / CUDA tensor; variances: same shape, a
    global ``(D,)`` or None (unit); windows: one window list for all dynamic streams or a list of
    window lists, one per stream; lengths: valid frames per utterance (3-D input).  Returns the
    static features of all streams side by side, same kind and rank as ``inputs``.
    """
    torch = _hip.torch_mod()
    is_np = not torch.is_tensor(inputs)
    dev = _hip.require_gpu(device if is_np or not inputs.is_cuda else inputs.device)
    m = torch.from_numpy(np.ascontiguousarray(_as_float(inputs))).to(dev) if is_np else inputs.to(dev).contiguous()
    two_d = m.dim() == 2
    if two_d:
        m = m[None]
    assert m.dim() == 3
    D = m.shape[-1]
    assert len(stream_sizes) == len(has_dynamic_features) and sum(stream_sizes) == D
    out_dtype = m.dtype
    if variances is None:
        v = None
    else:
        v = (variances if torch.is_tensor(variances) else
             torch.from_numpy(np.ascontiguousarray(_as_float(variances)))).to(dev).contiguous()
        m, v = _match_dtypes(m, v)
    if v is not None and v.dim() != 1:
        if two_d and v.dim() == 2:
            v = v[None]
        assert v.shape == m.shape
    per_stream = len(windows) > 0 and len(windows[0]) > 0 and isinstance(windows[0][0], (tuple, list))
    assert not per_stream or len(windows) == len(stream_sizes)
    streams, col = [], 0
    for k, (size, dyn) in enumerate(zip(stream_sizes, has_dynamic_features)):
        if dyn:
            w = windows[k] if per_stream else windows
            assert size % len(w) == 0
            streams.append((col, size // len(w), w))
        else:
            streams.append((col, size, None))
        col += size
    L = None
    if lengths is not None:
        L = torch.as_tensor(np.asarray(lengths) if not torch.is_tensor(lengths) else lengths).to(
            device=dev, dtype=torch.int32).contiguous()
    out, status = _hip.forward_streams(m, v, streams, L, algo=algo, want_status=check)
    if check:
        _hip.raise_on_status(status, out.shape[-1])
    if out.dtype != out_dtype:
        out = out.to(out_dtype)
    if two_d:
        out = out[0]
    return out.cpu().numpy() if is_np else out


def mlpg(mean_frames, variance_frames, windows):
    """Maximum Likelihood Parameter Generation, ``(T, D) -> (T, static_dim)``.

    Drop-in for ``nnmnkwii.paramgen.mlpg`` (paramgen/_mlpg.py:92-199): same
    arguments, same output dtype (that of ``mean_frames``), accepts per-frame
    ``(T, D)`` or global ``(D,)`` variances, raises ``AssertionError`` on a shape
    mismatch and ``numpy.linalg.LinAlgError`` ("k-th leading minor not positive
    definite") when a system is not positive definite.

    Windows: what the reference's ``build_win_mats`` accepts (paramgen/_mlpg.py:13-50) within the library's compile-time
    limits -- at most 8 windows, window extents ``l, u <= 4`` (half-bandwidth of ``P`` <= 8), at most 48 coefficients in
    total (csrc/common.h: kMaxWindows, kMaxExtent, kMaxCoef); beyond those the call raises ``HipExtensionError`` (the
    reference itself has no limit).  Which kernel a window set gets (``MLPG_HIP_ALGO_AUTO``; DESIGN.md "AUTO routing",
    measured in profiles/r05_auto_routing.json):

    * extents <= 1 (the usual static / delta / delta-delta set, also one or two windows): the fast kernels -- strip,
      wave-per-system, constant-coefficient, FIR -- at 0.4-0.5 of the HBM roofline for wide streams; batches of narrow
      streams (1-32 static dims: lf0, bap) ride the strip kernel with its lanes over several utterances;
    * extents of 2 (the reference's 5-tap test windows, tests/test_paramgen.py:21-26), up to three windows: the chunked
      kernel, which reads the inputs twice: 0.22 of the roofline;
    * more than three windows with an extent of 2, or extents of 3-4: the natural-order kernel, one lane per system and
      the factor through HBM: 0.04-0.07 of the roofline (correct, slow).
    """
    mean_frames = np.asarray(mean_frames)
    variance_frames = np.asarray(variance_frames)
    dtype = mean_frames.dtype
    T, D = mean_frames.shape
    if not (variance_frames.ndim == 1 and variance_frames.shape[0] == D):
        assert mean_frames.shape == variance_frames.shape
        variance_frames = variance_frames[None]
    y = mlpg_batch(mean_frames[None], variance_frames, windows)
    return y[0].astype(dtype, copy=False)


def mlpg_grad(mean_frames, variance_frames, windows, grad_output):
    """Gradient of MLPG w.r.t. the means, ``float32 (T, D)``.

    Drop-in for ``nnmnkwii.paramgen.mlpg_grad`` (paramgen/_mlpg.py:202-281).
    The reference solves a dense ``T x T`` right-hand side per (dim, window);
    the HIP kernel computes the same quantity in O(T):
    ``grad[:, w*sd+d] = tau_w * (W_w P_d^-1 o_d)``.  ``variance_frames`` ``(T, D)``, or a global ``(D,)`` (the reference
    leaves that broadcast to its caller, autograd/_impl/mlpg.py:196-197).
    """
    mean_frames = np.asarray(mean_frames)
    v = np.ascontiguousarray(_as_float(variance_frames))
    T, D = mean_frames.shape
    if v.ndim == 2:
        assert v.shape == (T, D)
        v = v[None]
    # numpy in -> numpy out through the host-memory entry point (mlpg_hip_backward_host: the library's short path -- no torch
    # tensor, one C call; the arithmetic runs in the variances' dtype class as before: float32 variances -> float32 inputs)
    go = np.ascontiguousarray(np.asarray(grad_output), dtype=v.dtype)
    assert go.shape == (T, D // len(windows))
    grad, status = _hip.backward_host(v, go[None], windows, D, out_dtype=np.float32)
    if status.any():
        _raise_host_status(status)
    return grad[0]


# registry of MLPG matrices handed out by unit_variance_mlpg_matrix, so that
# autograd.unit_variance_mlpg(R, means) can recover (windows, T) from R and run
# the banded O(T) kernels instead of a dense (T x nw*T) product.
import collections

_UV_REGISTRY = collections.OrderedDict()
_UV_REGISTRY_MAX = 8          # matrices remembered (each is a (T, nw*T) float32 array: 3 MB at T = 500)


def _fingerprint(R):
    """Cheap content key of an MLPG matrix: shape + a fixed sample of entries."""
    T, K = R.shape
    flat = R.reshape(-1)
    n = flat.shape[0]
    idx = (np.arange(97, dtype=np.int64) * 2654435761 + 12345) % max(n, 1)
    return (int(T), int(K)) + tuple(np.asarray(flat[idx], dtype=np.float32).tolist())


def unit_variance_mlpg_matrix(windows, T):
    """MLPG matrix ``R = (sum_w W~_w^T W_w)^-1 [W~_0^T ... W~_{nw-1}^T]``, float32 ``(T, nw*T)``.

    Drop-in for ``nnmnkwii.paramgen.unit_variance_mlpg_matrix``
    (paramgen/_mlpg.py:297-373).  Column ``w*T + t`` of ``R`` is the response to
    a unit mean at window ``w``, frame ``t``: the matrix is obtained with ONE
    unit-variance MLPG launch over ``nw*T`` one-hot "dimensions" instead of a
    dense banded inverse.  The returned array is a plain ndarray; it is also
    remembered so that ``autograd.unit_variance_mlpg`` can use the banded kernels.
    """
    torch = _hip.torch_mod()
    dev = _hip.require_gpu()
    nw = len(windows)
    K = nw * T
    E = torch.zeros((T, nw, K), dtype=torch.float64, device=dev)
    t = torch.arange(T, device=dev)
    for w in range(nw):
        E[t, w, w * T + t] = 1.0
    out, status = _hip.forward(E.view(1, T, nw * K), None, windows)
    _hip.raise_on_status(status, K)
    R = out[0].to(torch.float32).cpu().numpy()
    _UV_REGISTRY[_fingerprint(R)] = (tuple((int(l), int(u), tuple(np.asarray(c, dtype=np.float64).tolist()))
                                           for l, u, c in windows), int(T), R)
    while len(_UV_REGISTRY) > _UV_REGISTRY_MAX:      # least recently registered / looked up goes first
        _UV_REGISTRY.popitem(last=False)
    return R


def lookup_unit_variance_matrix(R_np_or_fp):
    """(windows, T, R) registered for this matrix, or None."""
    key = R_np_or_fp if isinstance(R_np_or_fp, tuple) else _fingerprint(R_np_or_fp)
    hit = _UV_REGISTRY.get(key)
    if hit is not None:
        _UV_REGISTRY.move_to_end(key)
    return hit


def reshape_means(means, static_dim):
    """``(T, D) -> (T*num_windows, static_dim)``; no-op if already reshaped
    (paramgen/_mlpg.py:376-405)."""
    T, D = means.shape
    if D == static_dim:
        return means
    return means.reshape(T, -1, static_dim).transpose(1, 0, 2).reshape(-1, static_dim)
