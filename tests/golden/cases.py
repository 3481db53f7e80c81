"""Deterministic input generators shared by make_golden.py and the tests.

Window sets are the ones the reference's own tests use
(/root/reference/tests/test_paramgen.py:4-28) plus the degenerate set that
triggers the ``precisions[-0:]`` slice quirk.
"""
import zlib

import numpy as np

WINDOW_SETS = {
    "static": [(0, 0, np.array([1.0]))],
    "std2": [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5]))],
    "std3": [
        (0, 0, np.array([1.0])),
        (1, 1, np.array([-0.5, 0.0, 0.5])),
        (1, 1, np.array([1.0, -2.0, 1.0])),
    ],
    "wide3": [
        (0, 0, np.array([1.0])),
        (2, 2, np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0),
        (2, 2, np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0),
    ],
    "zero2": [(0, 0, np.array([1.0])), (0, 0, np.array([2.0]))],
    "asym2": [(0, 0, np.array([1.0])), (1, 0, np.array([-1.0, 1.0]))],
}

_DT = {"f32": np.float32, "f64": np.float64}


def _seed(*parts):
    return zlib.crc32("/".join(str(p) for p in parts).encode()) & 0x7FFFFFFF


def rand_case(wname, dt, T, sd, salt=0):
    """means (T, D), per-frame variances (T, D), global variances (D,) in dtype dt."""
    nw = len(WINDOW_SETS[wname])
    rng = np.random.RandomState(_seed(wname, dt, T, sd, salt))
    D = nw * sd
    m = rng.randn(T, D).astype(_DT[dt])
    v = (rng.rand(T, D) + 0.1).astype(_DT[dt])
    vg = (rng.rand(D) + 0.1).astype(_DT[dt])
    return m, v, vg


def c2_utterance(b, T=1000, sd=60):
    """Utterance b of the BASELINE config-2 shaped batch: (T, 3*sd) float64."""
    rng = np.random.RandomState(1234 + b)
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    return m, v


def _track(rng, T, D):
    return np.cumsum(rng.randn(T, D), axis=0) * 0.3


def align_batch(name):
    """Zero-padded (N, Tx, D), (N, Ty, D) pairs of smooth tracks for the DTW aligner goldens.

    small   : content much shorter than the padding (outputs keep the padded length)
    grow    : full-length content (warping paths longer than the padding: outputs grow)
    xlonger : X padded longer than Y (the output dtype/length follow X)
    f32     : float32 inputs
    """
    rng = np.random.RandomState(_seed("align", name))
    N, D = 3, 4
    if name in ("small", "f32"):
        Tx, Ty, lo, hi = 40, 44, 12, 22
    elif name == "grow":
        Tx, Ty, lo, hi = 18, 18, 17, 19
    elif name == "xlonger":
        Tx, Ty, lo, hi = 36, 30, 14, 26
    else:
        raise KeyError(name)
    X = np.zeros((N, Tx, D))
    Y = np.zeros((N, Ty, D))
    for n in range(N):
        tx = int(rng.randint(lo, min(hi, Tx + 1)))
        ty = int(rng.randint(lo, min(hi, Ty + 1)))
        base = _track(rng, max(tx, ty) + 8, D)
        ix = np.sort(rng.choice(len(base), tx, replace=False))
        iy = np.sort(rng.choice(len(base), ty, replace=False))
        X[n, :tx] = base[ix] + 0.05 * rng.randn(tx, D)
        Y[n, :ty] = 0.8 * base[iy] + 0.3 + 0.05 * rng.randn(ty, D)
    if name == "f32":
        X, Y = X.astype(np.float32), Y.astype(np.float32)
    return X, Y


def gmm_joint_data(wname, sd, n=400, T=30):
    """(n, 2*D) joint source/target samples for a GMM fit and a (T, D) source utterance, D = nw*sd."""
    nw = len(WINDOW_SETS[wname])
    D = nw * sd
    rng = np.random.RandomState(_seed("gmm", wname, sd))
    centers = rng.randn(3, D) * 2.0
    which = rng.randint(0, 3, size=n)
    x = centers[which] + rng.randn(n, D)
    A = np.eye(D) * 0.7 + 0.1 * rng.randn(D, D)
    y = x @ A + 0.5 + 0.3 * rng.randn(n, D)
    src = centers[rng.randint(0, 3, size=T)] + rng.randn(T, D)
    return np.concatenate([x, y], axis=1), src


def c4_pairs(n, seed=4242):
    """n BASELINE config-4 sized pairs: smooth 25-dim tracks, T in [700, 900], zero-padded to 900 (float64)."""
    rng = np.random.RandomState(seed)
    X = np.zeros((n, 900, 25))
    Y = np.zeros((n, 900, 25))
    for k in range(n):
        a, b = rng.randint(700, 901, size=2)
        X[k, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1
        Y[k, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
    return X, Y
