"""Random soak of DTWAligner.transform (device-tensor route and host-pointer route) against the oracle's restatement of the
reference's transform (oracle/dtw.py: dtw_align): random batch sizes, lengths, dims, radius, zero-padded inputs, float32 /
float64.   usage: python tools/dbg/align_soak.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nnmnkwii_amd.preprocessing.alignment import DTWAligner  # noqa: E402
from oracle import dtw as OD  # noqa: E402


def soak(budget=40.0, seed=5):
    rng = np.random.RandomState(seed)
    t0 = time.time()
    n = 0
    bad = None
    while time.time() - t0 < budget and bad is None:
        N = int(rng.randint(1, 30))
        D = int(rng.randint(1, 12))
        Tx, Ty = int(rng.randint(4, 160)), int(rng.randint(4, 160))
        radius = int(rng.choice([1, 1, 2, 4]))
        dt = [np.float64, np.float32][rng.randint(2)]
        X = np.zeros((N, Tx, D), dtype=dt)
        Y = np.zeros((N, Ty, D), dtype=dt)
        for i in range(N):
            a, b = int(rng.randint(2, Tx + 1)), int(rng.randint(2, Ty + 1))
            X[i, :a] = (np.cumsum(rng.randn(a, D), 0) * 0.1 + 1.0).astype(dt)
            Y[i, :b] = (np.cumsum(rng.randn(b, D), 0) * 0.1 + 1.0).astype(dt)
        al = DTWAligner(radius=radius)
        host = rng.rand() < 0.5
        al._HOST_ENTRY_BYTES = 0 if host else (1 << 40)
        Xa, Ya = al.transform((X, Y))
        Xr, Yr, _, _ = OD.dtw_align(X, Y, radius)
        if Xa.shape != Xr.shape or Ya.shape != Yr.shape or not np.array_equal(Xa, Xr) or not np.array_equal(Ya, Yr):
            bad = (n, N, D, Tx, Ty, radius, dt.__name__, host, Xa.shape, Xr.shape)
        n += 1
    return n, bad


if __name__ == "__main__":
    print(soak(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0))
