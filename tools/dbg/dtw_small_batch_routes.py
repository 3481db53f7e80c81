import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd.preprocessing.alignment import DTWAligner
from oracle import dtw as OD
rng = np.random.RandomState(0)
for N in (1, 2, 4, 8):
    X = np.zeros((N, 900, 25)); Y = np.zeros((N, 900, 25))
    for n in range(N):
        a, b = 812 - 7 * n, 777 + 5 * n
        X[n, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1
        Y[n, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
    res = []
    for pairs in (0, 64):
        al = DTWAligner()
        al._HOST_ENTRY_PAIRS = pairs
        for _ in range(10):
            out = al.transform((X, Y))
        ts = []
        for _ in range(60):
            t0 = time.perf_counter(); out = al.transform((X, Y)); ts.append(time.perf_counter() - t0)
        res.append((np.median(ts) * 1e6, out))
    Xo, Yo = OD.dtw_align(X, Y)[:2]
    ok = all(np.array_equal(r[1][0], Xo) and np.array_equal(r[1][1], Yo) for r in res)
    print("N=%d pairs: framework route %.1f us, host route %.1f us; both equal to the oracle: %s" % (N, res[0][0], res[1][0], ok))
