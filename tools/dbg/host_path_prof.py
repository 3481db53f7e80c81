import sys, time
import numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from nnmnkwii_amd import paramgen as G
from tools.bench_paths import WINDOWS
B, T, sd = 256, 1000, 60
rng = np.random.RandomState(0)
mp, vp = _hip.pinned_empty((B, T, 3 * sd)), _hip.pinned_empty((B, T, 3 * sd))
mp[...] = rng.randn(B, T, 3 * sd); vp[...] = rng.rand(B, T, 3 * sd) + 0.1
G.mlpg_batch(mp, vp, WINDOWS)
for rep in range(3):
    t0 = time.perf_counter(); out, st = _hip.forward_host(mp, vp, WINDOWS); t1 = time.perf_counter()
    print("forward_host (pinned in, pageable out) %.2f ms" % ((t1 - t0) * 1e3))
