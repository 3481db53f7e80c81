"""CPU: the numpy model of the constant-coefficient kernel (tools/const_model.py, the executable specification of
csrc/mlpg_const_impl.h: factor once, second-order recurrences, chunks aligned to the utterance's end, one-step lag with
parked chunks, two-sweep fallback) against the oracle."""
import os
import sys

import numpy as np
import pytest

from cases import WINDOW_SETS, rand_case
from oracle import mlpg as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import const_model as C  # noqa: E402


def rel_err(y, ref):
    scale = np.abs(ref).max()
    return float(np.abs(y - ref).max() / (scale if scale else 1.0))


@pytest.mark.parametrize("wname", ["std3", "std2", "asym2"])
@pytest.mark.parametrize("T", [1, 2, 3, 4, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 66, 127, 129, 130, 200, 257, 600])
def test_model_vs_oracle_global_variances(wname, T):
    m, _, vg = rand_case(wname, "f64", T, 3, salt=11)
    for M, W in ((16, 8), (16, 2), (8, 3)):
        if T > 300 and W != 8:
            continue
        y, st = C.mlpg_const(m[None], vg, WINDOW_SETS[wname], M=M, W=W)
        assert not st.any()
        assert rel_err(y[0], O.mlpg(m, vg, WINDOW_SETS[wname])) < 1e-11, (wname, T, M, W)


def test_model_unit_variances_and_ragged_batch():
    rng = np.random.RandomState(3)
    means = rng.randn(5, 150, 6)
    lengths = np.array([150, 149, 33, 2, 1], dtype=np.int32)
    for var in (None, rng.rand(6) + 0.1):
        y, st = C.mlpg_const(means, var, WINDOW_SETS["std3"], lengths, M=16, W=2)
        yo, _, rc = O.mlpg_batch(means, np.ones(6) if var is None else var, WINDOW_SETS["std3"], lengths)
        assert rc == 0 and not st.any()
        assert rel_err(y, yo) < 1e-12
        for b, T in enumerate(lengths):
            assert not y[b, T:].any()


def test_model_parks_chunks_and_falls_back_to_two_sweeps():
    """Ordinary variances: the lowest chunks of a super-step wait for the next one (parked), nobody needs more.  Dynamic
    features 100x / 10000x tighter than the static ones: the factor converges slowly, chunks far from the super-step's
    bottom would have to wait too -- no slot -- and the utterance takes the two-sweep path.  Same result either way."""
    rng = np.random.RandomState(4)
    means = rng.randn(1, 700, 6)
    stats = {}
    var = rng.rand(6) + 0.1
    y, _ = C.mlpg_const(means, var, WINDOW_SETS["std3"], stats=stats)
    yo, _, rc = O.mlpg_batch(means, var, WINDOW_SETS["std3"])
    assert rc == 0 and rel_err(y, yo) < 1e-12
    assert stats["parked"] > 4 and stats["two_sweep"] == [False]
    stats = {}
    var = np.array([1.0, 2.0, 1e-2, 3e-2, 1e-4, 2e-4])
    y, _ = C.mlpg_const(means, var, WINDOW_SETS["std3"], stats=stats)
    yo, _, rc = O.mlpg_batch(means, var, WINDOW_SETS["std3"])
    assert rc == 0 and rel_err(y, yo) < 1e-9
    assert stats["i_s"] > 100 and stats["two_sweep"] == [True]
    # few slots: the same happens with ordinary variances
    stats = {}
    var = rng.rand(6) + 0.1
    y, _ = C.mlpg_const(means, var, WINDOW_SETS["std3"], stats=stats, slots=1)
    yo, _, rc = O.mlpg_batch(means, var, WINDOW_SETS["std3"])
    assert rc == 0 and rel_err(y, yo) < 1e-12 and stats["two_sweep"] == [True]


def test_model_flags_negative_global_variance():
    m, _, vg = rand_case("std3", "f64", 80, 2, salt=5)
    vg = vg.copy()
    vg[0] = -0.3   # static variance of dim 0
    _, st = C.mlpg_const(m[None], vg, WINDOW_SETS["std3"], M=16, W=2)
    _, so, rc = O.mlpg_batch(m[None], vg, WINDOW_SETS["std3"])
    assert rc != 0
    assert (st[0] != 0).tolist() == (so[0] != 0).tolist()


@pytest.mark.parametrize("T", [1, 2, 3, 7, 40, 130])
def test_model_backward_vs_oracle(T):
    rng = np.random.RandomState(T)
    sd = 2
    var = rng.rand(3 * sd) + 0.1
    go = rng.randn(T, sd)
    g = C.mlpg_const_backward(var, WINDOW_SETS["std3"], go[None], M=16, W=2)[0]
    gr = O.mlpg_grad(np.zeros((T, 3 * sd)), np.tile(var, (T, 1)), WINDOW_SETS["std3"], go)   # float32, as the reference
    assert np.abs(g - gr).max() < 5e-7 * max(1.0, np.abs(gr).max())
