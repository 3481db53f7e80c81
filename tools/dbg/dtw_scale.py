import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from nnmnkwii_amd import _hip
from tools.bench_paths import gpu_time
rng = np.random.RandomState(1234)
D = 25
for N in (32, 128, 256, 512, 1024):
    X = np.zeros((N, 900, D)); Y = np.zeros((N, 900, D))
    for n in range(N):
        a, b = rng.randint(700, 901, size=2)
        X[n, :a] = np.cumsum(rng.randn(a, D), 0) * 0.1; Y[n, :b] = np.cumsum(rng.randn(b, D), 0) * 0.1
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    lx, ly = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
    ms = gpu_time(lambda: _hip.fastdtw_l2(Xd, Yd, lx, ly, 1), steps=5)
    print(N, "pairs", round(ms, 3), "ms", round(N / ms * 1e3), "pairs/s", flush=True)
