"""Round-2 goldens, again from the REFERENCE ITSELF (tests/golden/mlpg_golden2.npz, align_golden2.npz).

Run in the build container only (needs /root/reference):

    bash oracle/build_reference.sh            # scratch build under /tmp/oracle_ref
    PYTHONPATH=/tmp/oracle_ref python tests/golden/make_golden2.py

* ``grad2/...``: the reference's O(T^2) ``paramgen.mlpg_grad`` (dense solve_banded per static dim and window,
  _mlpg.py:202-281) at T in {300, 1000, 2000}: the sizes where the HIP kernels run their M = 8 / 16 / 32
  instantiations and the strip kernel its multi-strip path.  float32 and float64 inputs.  Inputs are rebuilt by
  ``cases.rand_case`` (salt 11) and a seeded grad_output.
* ``uv3/...``: ``unit_variance_mlpg_matrix(std3, 500)`` row sums / a few rows (the whole 3 MB matrix is not stored;
  the config-3 test rebuilds the product from the oracle in float64).
* ``dtw4/...``: BASELINE config 4 sized pairs (T in [700, 900], 25-dim) through the reference's unmodified
  ``DTWAligner`` with its OWN ``dist = lambda x, y: norm(x - y)`` (numpy BLAS-order sum), the ``fastdtw`` import
  bound to the literal restatement oracle/dtw.py::fastdtw_py (the package itself is absent: parity of the path is
  conditional on that restatement, as oracle/dtw.py says).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from cases import WINDOW_SETS, c4_pairs, rand_case  # noqa: E402


def main():
    from oracle import dtw as OD

    def fastdtw(x, y, radius=1, dist=None):
        d, path = OD.fastdtw_py(x, y, radius=radius, dist=dist if dist is not None else OD.l2)
        return d, [(int(i), int(j)) for i, j in path]

    mod = types.ModuleType("fastdtw")
    mod.fastdtw = fastdtw
    sys.modules["fastdtw"] = mod

    import nnmnkwii
    assert "oracle_ref" in nnmnkwii.__file__ or "reference" in nnmnkwii.__file__, nnmnkwii.__file__
    from nnmnkwii import paramgen as G
    from nnmnkwii.preprocessing.alignment import DTWAligner

    out = {}
    windows = WINDOW_SETS["std3"]
    for T in (300, 1000, 2000):
        sd = 2
        for dt in ("f32", "f64"):
            m, v, _ = rand_case("std3", dt, T, sd, salt=11)
            go = np.random.RandomState(500 + T).randn(T, sd).astype(m.dtype)
            g = G.mlpg_grad(m, v, windows, go)
            assert g.dtype == np.float32
            out["grad2/std3-%s-T%d/g" % (dt, T)] = g
    for wname in ("std2", "asym2"):
        T, sd = 700, 3
        m, v, _ = rand_case(wname, "f64", T, sd, salt=11)
        go = np.random.RandomState(500 + T).randn(T, sd)
        out["grad2/%s-f64-T%d/g" % (wname, T)] = G.mlpg_grad(m, v, WINDOW_SETS[wname], go)

    R = G.unit_variance_mlpg_matrix(windows, 500)
    out["uv3/rowsum"] = R.sum(axis=1)
    out["uv3/rows"] = R[[0, 1, 2, 250, 497, 498, 499]]
    np.savez_compressed(os.path.join(HERE, "mlpg_golden2.npz"), **out)
    print("wrote mlpg_golden2.npz", len(out), "arrays")

    out = {}
    X, Y = c4_pairs(6)
    Xa, Ya = DTWAligner().transform((X, Y))     # the reference's own default dist (norm(x - y))
    out["dtw4/Xa"] = Xa
    out["dtw4/Ya"] = Ya
    # the reference's own custom-dist case (tests/test_preprocessing.py:496-501): dist = metrics.melcd
    from cases import align_batch
    from nnmnkwii.metrics import melcd
    for name in ("small", "grow"):
        X, Y = align_batch(name)
        Xa, Ya = DTWAligner(dist=melcd).transform((X, Y))
        out["dtw-melcd/%s/Xa" % name] = Xa
        out["dtw-melcd/%s/Ya" % name] = Ya
    X, Y = c4_pairs(2, seed=99)
    Xa, Ya = DTWAligner(dist=melcd).transform((X, Y))
    out["dtw-melcd/c4/Xa"] = Xa
    out["dtw-melcd/c4/Ya"] = Ya
    np.savez_compressed(os.path.join(HERE, "align_golden2.npz"), **out)
    print("wrote align_golden2.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
