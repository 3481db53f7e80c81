"""Generate tests/golden/align_golden.npz from the REFERENCE's own alignment / GMM code.

Run in the build container only (needs /root/reference):

    bash oracle/build_reference.sh            # scratch build under /tmp/oracle_ref
    PYTHONPATH=/tmp/oracle_ref python tests/golden/make_golden_align.py

What is the reference and what is not:

* ``nnmnkwii.preprocessing.alignment`` (DTWAligner, IterativeDTWAligner) and
  ``nnmnkwii.baseline.gmm`` (MLPGBase, MLPG) are imported UNMODIFIED from the
  scratch build of /root/reference and executed as they are.
* The third-party ``fastdtw`` package they import is not available (SURVEY.md
  8c).  Its place in ``sys.modules`` is taken by a stub whose ``fastdtw(x, y,
  radius, dist)`` is ``oracle.dtw.fastdtw_py`` -- the literal restatement of
  the published algorithm, calling the reference's own ``dist`` callable per
  cell.  The goldens therefore pin everything the reference does AROUND
  fastdtw (trim, gather, padding growth, GMM fit/convert loop, final gather)
  and are conditional on the restatement for the warping path itself
  ("parity unpinned" for the path, as oracle/dtw.py says).

sklearn's GaussianMixture draws from numpy's global RandomState when
``random_state`` is None (alignment.py:170-174), so each case seeds
``np.random.seed`` first; the tests do the same.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from cases import WINDOW_SETS, align_batch, gmm_joint_data  # noqa: E402


def _install_fastdtw_stub():
    from oracle import dtw as OD

    def fastdtw(x, y, radius=1, dist=None):
        d, path = OD.fastdtw_py(x, y, radius=radius, dist=dist if dist is not None else OD.l2)
        return d, [(int(i), int(j)) for i, j in path]

    mod = types.ModuleType("fastdtw")
    mod.fastdtw = fastdtw
    sys.modules["fastdtw"] = mod


def main():
    _install_fastdtw_stub()
    import nnmnkwii
    assert "oracle_ref" in nnmnkwii.__file__ or "reference" in nnmnkwii.__file__, nnmnkwii.__file__
    from nnmnkwii.baseline.gmm import MLPG
    from nnmnkwii.preprocessing.alignment import DTWAligner, IterativeDTWAligner
    from sklearn.mixture import GaussianMixture

    out = {}

    # --- DTWAligner.transform: ragged zero-padded pairs; "grow" forces a path longer than the padding
    for name in ("small", "grow", "xlonger", "f32"):
        X, Y = align_batch(name)
        Xa, Ya = DTWAligner().transform((X, Y))
        out["dtw/%s/Xa" % name] = Xa
        out["dtw/%s/Ya" % name] = Ya
    X, Y = align_batch("small")
    Xa, Ya = DTWAligner(radius=2).transform((X, Y))
    out["dtw/small-r2/Xa"] = Xa
    out["dtw/small-r2/Ya"] = Ya

    # --- baseline.gmm.MLPG.transform: joint GMM, static+delta features and static-only (frame-wise) features
    for wname in ("std2", "std3", "static"):
        windows = WINDOW_SETS[wname]
        sd = 3
        D = sd * len(windows)
        XYj, src = gmm_joint_data(wname, sd)
        gmm = GaussianMixture(n_components=3, covariance_type="full", random_state=7, max_iter=50).fit(XYj)
        key = "gmm/%s" % wname
        out[key + "/weights"] = gmm.weights_
        out[key + "/means"] = gmm.means_
        out[key + "/covariances"] = gmm.covariances_
        for swap in (False, True):
            for diff in (False, True):
                pg = MLPG(gmm, windows=windows, swap=swap, diff=diff)
                out[key + "/y-swap%d-diff%d" % (swap, diff)] = pg.transform(src)
        # static-only input to a delta-window model goes through the frame-wise conversion (gmm.py:216-217)
        if wname != "static":
            gs = GaussianMixture(n_components=2, covariance_type="full", random_state=3, max_iter=50).fit(XYj[:, [0, 1, 2, D, D + 1, D + 2]])
            out[key + "/s-weights"] = gs.weights_
            out[key + "/s-means"] = gs.means_
            out[key + "/s-covariances"] = gs.covariances_
            out[key + "/s-y"] = MLPG(gs, windows=[(0, 0, np.array([1.0]))]).transform(src[:, :3])

    # --- IterativeDTWAligner.transform
    for name, n_iter, ncomp in (("small", 1, 2), ("small", 2, 2), ("grow", 2, 3)):
        X, Y = align_batch(name)
        np.random.seed(1234)
        Xa, Ya = IterativeDTWAligner(n_iter=n_iter, n_components_gmm=ncomp, max_iter_gmm=20).transform((X, Y))
        out["iter/%s-it%d-k%d/Xa" % (name, n_iter, ncomp)] = Xa
        out["iter/%s-it%d-k%d/Ya" % (name, n_iter, ncomp)] = Ya

    path = os.path.join(HERE, "align_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
