"""CPU (-m "not gpu"): tools/walk_model.py -- the executable specification of the strip kernel's WALK form (one workgroup walks an
utterance's strips in order: exact top-down carry, one strip of look-ahead with the 2^-66 bound, the last strip exact) -- against the
oracle, and that it REJECTS utterances whose strips are coupled too tightly for the look-ahead (they go to the general route)."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import mlpg as O
from tools import walk_model as WM

W3 = WINDOW_SETS["std3"]


@pytest.mark.parametrize("T,sd", [(1, 2), (2, 2), (3, 2), (17, 2), (63, 3), (64, 3), (65, 3), (130, 2), (257, 3), (999, 5), (1000, 4)])
def test_walk_model_matches_the_oracle(T, sd):
    rng = np.random.RandomState(T * 7 + sd)
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    damps = []
    y, bad, accepted = WM.mlpg_walk(m, v, W3, stats=damps)
    yo = O.mlpg(m, v, W3)
    assert accepted and not bad.any()
    assert np.abs(y - yo).max() <= 1e-12 * np.abs(yo).max()
    assert all(d < 2.0 ** -66 for d in damps)


def test_walk_model_ragged_and_padding():
    rng = np.random.RandomState(5)
    m = rng.randn(300, 6)
    v = rng.rand(300, 6) + 0.1
    y, bad, accepted = WM.mlpg_walk(m, v, W3, T=201)
    yo = O.mlpg(m[:201], v[:201], W3)
    assert accepted and np.abs(y[:201] - yo).max() <= 1e-12 * np.abs(yo).max() and not y[201:].any()


@pytest.mark.parametrize("s1,s2", [(1e-1, 1e-2), (1e-2, 1e-3)])
def test_walk_model_rejects_tight_dynamic_variances(s1, s2):
    """Dynamic variances 10 x / 100 x (and more) tighter than the static ones: the trajectory is smooth over many strips, one strip of
    look-ahead is not enough -- the bound says so, and using the result anyway would be off by 1e-8 .. 1e-3."""
    rng = np.random.RandomState(11)
    T, sd = 1000, 3
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    v[:, sd:2 * sd] *= s1
    v[:, 2 * sd:] *= s2
    damps = []
    y, bad, accepted = WM.mlpg_walk(m, v, W3, stats=damps)
    assert not accepted and max(damps) > 1e-12


def test_walk_model_failing_pivot_is_reported():
    rng = np.random.RandomState(3)
    m = rng.randn(200, 6)
    v = rng.rand(200, 6) + 0.1
    v[77, 1] = -1e-6
    y, bad, accepted = WM.mlpg_walk(m, v, W3)
    assert bad.tolist() == [False, True]
