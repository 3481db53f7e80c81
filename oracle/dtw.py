"""DTW oracle (CPU) -- TEST INFRASTRUCTURE ONLY, never imported by the product.

PARITY UNPINNED: the alignment arithmetic of the reference's ``DTWAligner``
(/root/reference/nnmnkwii/preprocessing/alignment.py:9-76) lives in the
third-party PyPI package ``fastdtw`` (slaypni/fastdtw; unpinned in the
reference's setup.py:139; not under /root/reference; not installed here).  The
reference's tests pin shapes and a norm inequality only
(tests/test_preprocessing.py:441-501).  This module restates

* ``fastdtw_py``  -- the published algorithm with the semantics of upstream's
  pure-Python ``fastdtw/fastdtw.py`` (dict/set based, literal; small cases only),
* ``fastdtw``     -- the same in C (``dtw_oracle.c``), used at real sizes,
* ``trim_zeros_frames`` (preprocessing/generic.py:291-332, trim="b"),
* ``dtw_align``   -- ``DTWAligner.transform`` (alignment.py:40-76).

``fastdtw_py`` and ``fastdtw`` are checked against each other in
tests/test_oracle.py; the tie rule (first minimum of up, left, diagonal) is the
one documented in ``dtw_oracle.c``.
"""
import ctypes
from collections import defaultdict

import numpy as np

from .mlpg import _ptr, lib as _lib


def l2(a, b):
    """sqrt(sum (a-b)^2), sequential float64 sum, separate multiply and add."""
    acc = 0.0
    for u, v in zip(a, b):
        diff = float(u) - float(v)
        acc = acc + diff * diff
    return float(np.sqrt(acc))


# ------------------------------------------------------------ literal python

def _min_first(a, b, c):
    """upstream's pure-Python rule: Python's min keeps the FIRST minimal candidate (up, left, diagonal)"""
    return min(a, b, c, key=lambda t: t[0])


def _min_diag_last(a, b, c):
    """the strict-less chain recalled for upstream's compiled _fastdtw (UNVERIFIED): up only if it
    beats both others, else left only if it beats the diagonal, else the diagonal"""
    if a[0] < b[0] and a[0] < c[0]:
        return a
    return b if b[0] < c[0] else c


def _dtw_py(x, y, window, dist, pick=_min_first):
    len_x, len_y = len(x), len(y)
    if window is None:
        window = [(i, j) for i in range(len_x) for j in range(len_y)]
    window = ((i + 1, j + 1) for i, j in window)
    D = defaultdict(lambda: (float("inf"),))
    D[0, 0] = (0, 0, 0)
    for i, j in window:
        dt = dist(x[i - 1], y[j - 1])
        D[i, j] = pick(
            (D[i - 1, j][0] + dt, i - 1, j),
            (D[i, j - 1][0] + dt, i, j - 1),
            (D[i - 1, j - 1][0] + dt, i - 1, j - 1),
        )
    path = []
    i, j = len_x, len_y
    while not (i == j == 0):
        path.append((i - 1, j - 1))
        i, j = D[i, j][1], D[i, j][2]
    path.reverse()
    return D[len_x, len_y][0], path


def _reduce_by_half(x):
    return [(x[i] + x[1 + i]) / 2 for i in range(0, len(x) - len(x) % 2, 2)]


def _expand_window(path, len_x, len_y, radius):
    path_ = set(path)
    for i, j in path:
        for a in range(-radius, radius + 1):
            for b in range(-radius, radius + 1):
                path_.add((i + a, j + b))
    window_ = set()
    for i, j in path_:
        for a, b in ((i * 2, j * 2), (i * 2, j * 2 + 1), (i * 2 + 1, j * 2), (i * 2 + 1, j * 2 + 1)):
            window_.add((a, b))
    window = []
    start_j = 0
    for i in range(0, len_x):
        new_start_j = None
        for j in range(start_j, len_y):
            if (i, j) in window_:
                window.append((i, j))
                if new_start_j is None:
                    new_start_j = j
            elif new_start_j is not None:
                break
        start_j = new_start_j
    return window


def _fastdtw_py(x, y, radius, dist, pick=_min_first):
    min_time_size = radius + 2
    if len(x) < min_time_size or len(y) < min_time_size:
        return _dtw_py(x, y, None, dist, pick)
    x_shrinked = _reduce_by_half(x)
    y_shrinked = _reduce_by_half(y)
    _, path = _fastdtw_py(x_shrinked, y_shrinked, radius, dist, pick)
    window = _expand_window(path, len(x), len(y), radius)
    return _dtw_py(x, y, window, dist, pick)


def fastdtw_py(x, y, radius=1, dist=l2, tie=0):
    """Literal pure-Python restatement (small cases only).  tie: 0 = first minimum (pure Python), 1 = diagonal last."""
    x = np.asanyarray(x, dtype="float")
    y = np.asanyarray(y, dtype="float")
    return _fastdtw_py(x, y, radius, dist, _min_first if tie == 0 else _min_diag_last)


# ------------------------------------------------------------------- C path

def fastdtw(x, y, radius=1, tie=0):
    """C restatement with the L2 local cost. Returns (distance, path (n, 2) int32).  tie as fastdtw_py."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
        y = y[:, None]
    tx, D = x.shape
    ty = y.shape[0]
    pi = np.zeros(tx + ty, dtype=np.int32)
    pj = np.zeros(tx + ty, dtype=np.int32)
    cost = ctypes.c_double(0.0)
    L = _lib()
    L.oracle_fastdtw_l2_tie.restype = ctypes.c_long
    L.oracle_fastdtw_l2_tie.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                        ctypes.c_long, ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.POINTER(ctypes.c_double)]
    n = L.oracle_fastdtw_l2_tie(_ptr(x), tx, _ptr(y), ty, D, radius, int(tie), _ptr(pi), _ptr(pj), ctypes.byref(cost))
    if n < 0:
        raise RuntimeError("fastdtw oracle: unreachable window")
    return cost.value, np.stack([pi[:n], pj[:n]], axis=1)


def trim_zeros_frames(x, eps=1e-7):
    """preprocessing/generic.py:291-332 with trim="b": drop trailing frames with sum|x| < eps."""
    s = np.sum(np.abs(x), axis=1)
    s[s < eps] = 0.0
    end = len(np.trim_zeros(s, trim="b")) - len(x)
    return x if end == 0 else x[:end]


def dtw_align(X, Y, radius=1, use_c=True):
    """DTWAligner(radius=radius).transform((X, Y)) restated (alignment.py:40-76).

    Returns (X_aligned, Y_aligned, paths, dists).
    """
    assert X.ndim == 3 and Y.ndim == 3
    longer = X if X.shape[1] > Y.shape[1] else Y
    X_al = np.zeros_like(longer)
    Y_al = np.zeros_like(longer)
    paths, dists = [], []
    for idx, (x, y) in enumerate(zip(X, Y)):
        x, y = trim_zeros_frames(x), trim_zeros_frames(y)
        if use_c:
            dist, path = fastdtw(x, y, radius=radius)
        else:
            dist, path = fastdtw_py(x, y, radius=radius)
            path = np.asarray(path, dtype=np.int32)
        dist /= len(x) + len(y)
        x, y = x[path[:, 0]], y[path[:, 1]]
        max_len = max(len(x), len(y))
        if max_len > X_al.shape[1] or max_len > Y_al.shape[1]:
            pad_size = max(max_len - X_al.shape[1], max_len > Y_al.shape[1])
            X_al = np.pad(X_al, [(0, 0), (0, pad_size), (0, 0)], mode="constant", constant_values=0)
            Y_al = np.pad(Y_al, [(0, 0), (0, pad_size), (0, 0)], mode="constant", constant_values=0)
        X_al[idx][: len(x)] = x
        Y_al[idx][: len(y)] = y
        paths.append(path)
        dists.append(dist)
    return X_al, Y_al, paths, dists
