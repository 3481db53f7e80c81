"""Round-3 golden: 64 BASELINE config-4 sized pairs through the REFERENCE's DTWAligner with its own
``dist = lambda x, y: norm(x - y)`` (tests/golden/dtw_paths64.npz).

Run in the build container only (needs /root/reference):

    bash oracle/build_reference.sh            # scratch build under /tmp/oracle_ref
    PYTHONPATH=/tmp/oracle_ref python tests/golden/make_golden3.py

The reference's unmodified ``preprocessing/alignment.py`` is imported and run; its ``fastdtw`` import is bound to the
literal restatement ``oracle/dtw.py::fastdtw_py`` (the PyPI package is absent: parity of the path is conditional on
that restatement, as oracle/dtw.py says), which here RECORDS what the reference asked for and what came back: the
trimmed lengths of each pair, the warping path and the distance -- all computed with the callable the reference
passed, i.e. numpy's ``norm`` (a BLAS ``dot``, not a sequential sum).  Aligned feature arrays of 64 pairs would be
40 MB; the paths (int16) are 0.3 MB and say the same thing: ``X_aligned[n] = X[n][path_i]``.
What this pins: that the local cost evaluated in another summation order (sequential in the C oracle and in the HIP
kernel) never changes a decision of the DP on config-4 data -- 64 pairs, about a million window cells.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from cases import c4_pairs  # noqa: E402

N_PAIRS, SEED = 64, 64


def main():
    from oracle import dtw as OD
    calls = []

    def fastdtw(x, y, radius=1, dist=None):
        d, path = OD.fastdtw_py(x, y, radius=radius, dist=dist if dist is not None else OD.l2)
        path = [(int(i), int(j)) for i, j in path]
        calls.append((len(x), len(y), float(d), np.asarray(path, dtype=np.int16)))
        return d, path

    mod = types.ModuleType("fastdtw")
    mod.fastdtw = fastdtw
    sys.modules["fastdtw"] = mod
    import nnmnkwii
    assert "oracle_ref" in nnmnkwii.__file__ or "reference" in nnmnkwii.__file__, nnmnkwii.__file__
    from nnmnkwii.preprocessing.alignment import DTWAligner

    X, Y = c4_pairs(N_PAIRS, seed=SEED)
    Xa, Ya = DTWAligner().transform((X, Y))     # the reference's own default dist: norm(x - y)
    assert len(calls) == N_PAIRS
    L = max(len(c[3]) for c in calls)
    paths = np.full((N_PAIRS, L, 2), -1, dtype=np.int16)
    for n, c in enumerate(calls):
        paths[n, : len(c[3])] = c[3]
        # the reference's gather is indexing only: its outputs are X[n][path_i], Y[n][path_j]
        k = len(c[3])
        assert np.array_equal(Xa[n, :k], X[n][c[3][:, 0]]) and np.array_equal(Ya[n, :k], Y[n][c[3][:, 1]])
    out = {
        "lenx": np.array([c[0] for c in calls], dtype=np.int32),
        "leny": np.array([c[1] for c in calls], dtype=np.int32),
        "dist": np.array([c[2] for c in calls], dtype=np.float64),
        "plen": np.array([len(c[3]) for c in calls], dtype=np.int32),
        "paths": paths,
        "T_out": np.array([Xa.shape[1]], dtype=np.int32),
    }
    np.savez_compressed(os.path.join(HERE, "dtw_paths64.npz"), **out)
    print("wrote dtw_paths64.npz:", {k: v.shape for k, v in out.items()}, "cells on the paths:", int(out["plen"].sum()))


if __name__ == "__main__":
    main()
