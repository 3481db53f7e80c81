"""(Needs a library built with MLPG_HIP_EXTRA_FLAGS=-DMLPG_DTW_MEASURE: the switches below are not in the shipping build.)
fastdtw kernel time against the number of config-4 pairs (MLPG_HIP_DTW_FORCE=1: 256 threads + retry launch, 2: 512 threads)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nnmnkwii_amd import _hip
from bench_paths import gpu_time
rng = np.random.RandomState(1234)
N = 128
X = np.zeros((N, 900, 25)); Y = np.zeros((N, 900, 25))
for n in range(N):
    a, b = rng.randint(700, 901, size=2)
    X[n, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1; Y[n, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
X8, Y8 = torch.from_numpy(X).cuda().repeat(16, 1, 1).contiguous(), torch.from_numpy(Y).cuda().repeat(16, 1, 1).contiguous()
out = {}
for n in (1, 128, 256, 512, 768, 1024, 2048):
    Xd, Yd = X8[:n].contiguous(), Y8[:n].contiguous()
    lx, ly = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
    out[n] = round(gpu_time(lambda: _hip.fastdtw_l2(Xd, Yd, lx, ly, 1), steps=10), 4)
print("force", os.environ.get("MLPG_HIP_DTW_FORCE", "0"), out)
