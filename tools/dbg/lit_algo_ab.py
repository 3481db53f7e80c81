import sys, time, numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import paramgen as G, _hip
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
for T, sd in ((1000, 60), (2000, 60), (500, 60)):
    m, v = rng.randn(1, T, 3 * sd), rng.rand(1, T, 3 * sd) + 0.1
    res = []
    for rep in range(3):
        for algo, nm in ((0, "auto"), (2, "wave"), (3, "strip")):
            for _ in range(20): G.mlpg_batch(m, v, W, algo=algo)
            ts = []
            for _ in range(200):
                t0 = time.perf_counter(); G.mlpg_batch(m, v, W, algo=algo); ts.append(time.perf_counter() - t0)
            res.append("%s %.1f" % (nm, np.median(ts) * 1e6))
    print("T=%d sd=%d:" % (T, sd), " | ".join(res))
