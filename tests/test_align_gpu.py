"""IterativeDTWAligner and baseline.gmm.MLPG against goldens produced by the
reference's OWN alignment.py / baseline/gmm.py (tests/golden/make_golden_align.py; the fastdtw import of
the reference is satisfied by the oracle's literal restatement, so warping paths are pinned to that)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from cases import WINDOW_SETS, align_batch, gmm_joint_data  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "align_golden.npz"))


@pytest.mark.parametrize("name,radius", [("small", 1), ("grow", 1), ("xlonger", 1), ("f32", 1), ("small-r2", 2)])
def test_dtw_aligner_matches_reference_transform(golden, name, radius):
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    X, Y = align_batch(name.split("-")[0])
    Xa, Ya = DTWAligner(radius=radius).transform((X, Y))
    gx, gy = golden["dtw/%s/Xa" % name], golden["dtw/%s/Ya" % name]
    assert Xa.shape == gx.shape and Ya.shape == gy.shape
    assert Xa.dtype == gx.dtype and Ya.dtype == gy.dtype
    np.testing.assert_array_equal(Xa, gx)      # gathers of input rows: bit-exact
    np.testing.assert_array_equal(Ya, gy)


def _gmm_from(golden, key, prefix=""):
    from sklearn.mixture import GaussianMixture
    w, mu, cov = golden[key + "/%sweights" % prefix], golden[key + "/%smeans" % prefix], golden[key + "/%scovariances" % prefix]
    g = GaussianMixture(n_components=len(w), covariance_type="full")
    g.weights_, g.means_, g.covariances_ = w, mu, cov
    return g


@pytest.mark.parametrize("wname", ["std2", "std3", "static"])
def test_gmm_mlpg_transform_matches_reference(golden, wname):
    from nnmnkwii_amd.baseline.gmm import MLPG
    windows = WINDOW_SETS[wname]
    _, src = gmm_joint_data(wname, 3)
    gmm = _gmm_from(golden, "gmm/%s" % wname)
    for swap in (False, True):
        for diff in (False, True):
            y = MLPG(gmm, windows=windows, swap=swap, diff=diff).transform(src)
            ref = golden["gmm/%s/y-swap%d-diff%d" % (wname, swap, diff)]
            assert y.shape == ref.shape and y.dtype == ref.dtype
            # float64 trajectory: tolerance of the north star (1e-4 relative), observed ~1e-12
            np.testing.assert_allclose(y, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    if wname != "static":
        gs = _gmm_from(golden, "gmm/%s" % wname, "s-")
        y = MLPG(gs, windows=[(0, 0, np.array([1.0]))]).transform(src[:, :3])
        np.testing.assert_allclose(y, golden["gmm/%s/s-y" % wname], rtol=1e-10, atol=1e-12)
        # single frame (1-D input): MLPGBase.transform -> _transform_frame (gmm.py:94-95)
        from nnmnkwii_amd.baseline.gmm import MLPGBase
        y0 = MLPGBase(gs).transform(src[0, :3])
        np.testing.assert_allclose(y0, golden["gmm/%s/s-y" % wname][0], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("name,n_iter,ncomp", [("small", 1, 2), ("small", 2, 2), ("grow", 2, 3)])
def test_iterative_dtw_aligner_matches_reference(golden, name, n_iter, ncomp):
    from nnmnkwii_amd.preprocessing.alignment import IterativeDTWAligner
    X, Y = align_batch(name)
    X0, Y0 = X.copy(), Y.copy()
    np.random.seed(1234)            # sklearn's GaussianMixture(random_state=None) draws from numpy's global state
    Xa, Ya = IterativeDTWAligner(n_iter=n_iter, n_components_gmm=ncomp, max_iter_gmm=20).transform((X, Y))
    key = "iter/%s-it%d-k%d" % (name, n_iter, ncomp)
    assert Xa.shape == golden[key + "/Xa"].shape
    np.testing.assert_array_equal(Xa, golden[key + "/Xa"])
    np.testing.assert_array_equal(Ya, golden[key + "/Ya"])
    np.testing.assert_array_equal(X, X0)       # inputs are not modified
    np.testing.assert_array_equal(Y, Y0)


def test_iterative_aligner_reduces_distance_and_keeps_shapes():
    """The reference's own test for this class (tests/test_preprocessing.py:460-501): shapes agree and
    the aligned pair is closer than the unaligned one."""
    from nnmnkwii_amd.preprocessing.alignment import IterativeDTWAligner
    X, Y = align_batch("xlonger")
    np.random.seed(0)
    Xa, Ya = IterativeDTWAligner(n_iter=2, n_components_gmm=2, max_iter_gmm=10).transform((X, Y))
    assert Xa.shape == Ya.shape
    T = min(X.shape[1], Y.shape[1])
    assert np.linalg.norm(Xa - Ya) < np.linalg.norm(X[:, :T] - Y[:, :T])
