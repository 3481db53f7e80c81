"""Wall time of the host-memory calls (numpy -> numpy) over a ladder of sizes, for A/B runs of the library's host-path switches
(MLPG_HIP_HOST_DIRECT_KB, MLPG_HIP_HOST_SMALL_MB, MLPG_HIP_HOST_HELPERS: read once per process -- one process per setting).
usage: python tools/dbg/host_sizes.py [label] [big]      (big: also the whole config-2 batch, 737 MB in)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
label = sys.argv[1] if len(sys.argv) > 1 else ""
big = len(sys.argv) > 2
rng = np.random.RandomState(0)


def wall(fn, n, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6, min(ts) * 1e6


print("== %s  (MLPG_HIP_HOST_DIRECT_KB=%s MLPG_HIP_HOST_SMALL_MB=%s)" % (label, os.environ.get("MLPG_HIP_HOST_DIRECT_KB"), os.environ.get("MLPG_HIP_HOST_SMALL_MB")))
cases = [("c1 T=100 sd=2 f64", 1, 100, 2, np.float64, 400), ("T=500 sd=60 f64", 1, 500, 60, np.float64, 200), ("c2utt T=1000 sd=60 f32", 1, 1000, 60, np.float32, 200),
         ("c2utt T=1000 sd=60 f64", 1, 1000, 60, np.float64, 200), ("T=2000 sd=60 f64", 1, 2000, 60, np.float64, 100), ("B=4 T=1000 sd=60 f64", 4, 1000, 60, np.float64, 50),
         ("B=8 T=1000 sd=60 f64 (23 MB)", 8, 1000, 60, np.float64, 30), ("B=20 T=1000 sd=60 f64 (58 MB)", 20, 1000, 60, np.float64, 20)]
if big:
    cases.append(("config 2: B=256 T=1000 sd=60 f64 (737 MB)", 256, 1000, 60, np.float64, 5))
for name, B, T, sd, dt, n in cases:
    m = rng.randn(B, T, 3 * sd).astype(dt)
    v = (rng.rand(B, T, 3 * sd) + 0.1).astype(dt)
    go = rng.randn(B, T, sd).astype(dt)
    f = wall(lambda: G.mlpg_batch(m, v, W), n)
    # fresh arrays per call (a numpy temporary: pages the runtime has never seen)
    ts = []
    for _ in range(min(n, 20)):
        m2, v2 = m.copy(), v.copy()
        t0 = time.perf_counter()
        G.mlpg_batch(m2, v2, W)
        ts.append(time.perf_counter() - t0)
    fresh = np.median(ts) * 1e6
    # distinct arrays, each handed to the library ONCE (a list of utterances, as in the reference's loop): pages the runtime has never
    # seen, sources cold in the CPU's caches, and a result array that is kept
    K = int(max(4, min(48, (1 << 30) // (2 * m.nbytes))))
    pool = [(rng.randn(B, T, 3 * sd).astype(dt), (rng.rand(B, T, 3 * sd) + 0.1).astype(dt)) for _ in range(K)]
    keep = []
    ts = []
    for mm, vv in pool:
        t0 = time.perf_counter()
        keep.append(G.mlpg_batch(mm, vv, W))
        ts.append(time.perf_counter() - t0)
    once = np.median(ts) * 1e6
    del pool, keep
    b = wall(lambda: _hip.backward_host(v, go, W, 3 * sd, out_dtype=np.float32), n)
    u = wall(lambda: G.mlpg_batch(m, None, W), n)
    print("%-44s forward %9.1f us (min %9.1f; fresh arrays %9.1f; DISTINCT arrays, one call each %9.1f)   backward %9.1f us (min %9.1f)   forward unit variances %9.1f us"
          % (name, f[0], f[1], fresh, once, b[0], b[1], u[0]))
