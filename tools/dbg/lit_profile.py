"""Where the host time of ONE literal paramgen.mlpg(numpy) call goes (BASELINE config 1: T = 100, 2 static dims; one
config-2 utterance: T = 1000, 60 dims): cProfile of 2000 / 500 calls, and the C call alone (MLPG_HIP_HOST_TRACE-free)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(1234)
for T, sd, n in ((100, 2, 2000), (1000, 60, 500)):
    m = rng.randn(T, 3 * sd)
    v = rng.rand(T, 3 * sd) + 0.1
    for _ in range(20):
        G.mlpg(m, v, W)
    t0 = time.perf_counter()
    for _ in range(n):
        G.mlpg(m, v, W)
    per = (time.perf_counter() - t0) / n * 1e6
    # the C entry point alone on prepared arguments
    m3, v3 = m[None].copy(), v[None].copy()
    t0 = time.perf_counter()
    for _ in range(n):
        _hip.forward_host(m3, v3, W)
    per_c = (time.perf_counter() - t0) / n * 1e6
    print("T=%d sd=%d: paramgen.mlpg %.1f us per call; _hip.forward_host %.1f us per call" % (T, sd, per, per_c))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        G.mlpg(m, v, W)
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(14)
