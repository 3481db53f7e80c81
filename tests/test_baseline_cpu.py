"""baseline.gmm: host logic (attributes, swap/diff algebra: no GPU) and the frame-wise conversion (MLPGBase: one
mlpg_hip_gmm_convert launch; -m gpu) against the reference's baseline/gmm.py on the goldens it produced
(make_golden_align.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from cases import WINDOW_SETS, gmm_joint_data  # noqa: E402


def _gmm(golden, key, prefix=""):
    from sklearn.mixture import GaussianMixture
    w = golden[key + "/%sweights" % prefix]
    g = GaussianMixture(n_components=len(w), covariance_type="full")
    g.weights_, g.means_, g.covariances_ = w, golden[key + "/%smeans" % prefix], golden[key + "/%scovariances" % prefix]
    return g


@pytest.mark.gpu
def test_framewise_conversion_matches_reference():
    from nnmnkwii_amd.baseline.gmm import MLPG, MLPGBase
    golden = np.load(os.path.join(HERE, "golden", "align_golden.npz"))
    _, src = gmm_joint_data("static", 3)
    gmm = _gmm(golden, "gmm/static")
    for swap in (False, True):
        for diff in (False, True):
            y = MLPG(gmm, windows=WINDOW_SETS["static"], swap=swap, diff=diff).transform(src)
            np.testing.assert_allclose(y, golden["gmm/static/y-swap%d-diff%d" % (swap, diff)], rtol=1e-10, atol=1e-12)
    for wname in ("std2", "std3"):
        _, s = gmm_joint_data(wname, 3)
        gs = _gmm(golden, "gmm/%s" % wname, "s-")
        ref = golden["gmm/%s/s-y" % wname]
        np.testing.assert_allclose(MLPGBase(gs).transform(s[:, :3]), ref, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(MLPGBase(gs).transform(s[0, :3]), ref[0], rtol=1e-10, atol=1e-12)
        # output dtype follows the input (gmm.py:89 zeros_like)
        assert MLPGBase(gs).transform(s[:, :3].astype(np.float32)).dtype == np.float32
        # the batched form used by IterativeDTWAligner
        parts = MLPGBase(gs).transform_batch([s[:7, :3], s[7:, :3]])
        np.testing.assert_allclose(np.concatenate(parts), ref, rtol=1e-10, atol=1e-12)


def test_attributes_and_swap_diff_algebra():
    from nnmnkwii_amd.baseline.gmm import MLPGBase
    golden = np.load(os.path.join(HERE, "golden", "align_golden.npz"))
    gmm = _gmm(golden, "gmm/std2")
    a = MLPGBase(gmm)
    D = gmm.means_.shape[1] // 2
    assert a.num_mixtures == 3 and a.src_means.shape == (3, D) and a.covarYX.shape == (3, D, D)
    s = MLPGBase(gmm, swap=True)
    np.testing.assert_array_equal(s.src_means, a.tgt_means)
    np.testing.assert_array_equal(s.covarXY, a.covarYX)
    d = MLPGBase(gmm, diff=True)
    np.testing.assert_allclose(d.tgt_means, a.tgt_means - a.src_means)
    np.testing.assert_allclose(d.covarYX, d.covarXY.transpose(0, 2, 1))
    with pytest.raises(AssertionError):
        from sklearn.mixture import GaussianMixture
        g = GaussianMixture(n_components=2, covariance_type="diag")
        g.means_ = np.zeros((2, 4))
        MLPGBase(g)
