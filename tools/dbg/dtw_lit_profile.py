"""Where the host time of ONE DTWAligner.transform((X, Y)) call on a single pair goes (812 x 777 frames, 25 dims): cProfile of 300 calls,
and the C entry point alone (mlpg_hip_fastdtw_host) on prepared arguments."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd.preprocessing.alignment import DTWAligner  # noqa: E402

rng = np.random.RandomState(0)
a, b = 812, 777
X = np.zeros((1, 900, 25))
Y = np.zeros((1, 900, 25))
X[0, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1
Y[0, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
al = DTWAligner()
for _ in range(20):
    al.transform((X, Y))
n = 300
t0 = time.perf_counter()
for _ in range(n):
    al.transform((X, Y))
print("DTWAligner.transform: %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
t0 = time.perf_counter()
for _ in range(n):
    _hip.fastdtw_host(X, Y)
print("_hip.fastdtw_host (the C call + its numpy outputs): %.1f us per call" % ((time.perf_counter() - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    al.transform((X, Y))
pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("tottime").print_stats(16)
