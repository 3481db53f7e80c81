"""(Needs a library built with MLPG_HIP_EXTRA_FLAGS=-DMLPG_DTW_MEASURE: the switches below are not in the shipping build.)
1024 config-4 pairs: time of the two-launch form, and how many pairs the optimistic first launch gives up on
(run with MLPG_HIP_DTW_FIRST_LAUNCH_ONLY=1 for the latter: path_len == -1 stays visible)."""
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
Xd, Yd = torch.from_numpy(X).cuda().repeat(8, 1, 1).contiguous(), torch.from_numpy(Y).cuda().repeat(8, 1, 1).contiguous()
lx, ly = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
pi, pj, pl, c = _hip.fastdtw_l2(Xd, Yd, lx, ly, 1)
pl = pl.cpu().numpy()
print("pairs", len(pl), "marked -1:", int((pl == -1).sum()), "failed 0:", int((pl == 0).sum()))
print("ms", gpu_time(lambda: _hip.fastdtw_l2(Xd, Yd, lx, ly, 1), steps=10))
