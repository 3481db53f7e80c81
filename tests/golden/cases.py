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
