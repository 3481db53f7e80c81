"""Phase cycle counters of the fastdtw kernel (library built with -DMLPG_DTW_TIMING, see tools/gpurun/README.md);
argument: number of pairs (default 16)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from nnmnkwii_amd import _hip
rng = np.random.RandomState(1234)
N, D = 128, 25
X = np.zeros((N, 900, D)); Y = np.zeros((N, 900, D))
for n in range(N):
    a, b = rng.randint(700, 901, size=2)
    X[n, :a] = np.cumsum(rng.randn(a, D), 0) * 0.1; Y[n, :b] = np.cumsum(rng.randn(b, D), 0) * 0.1
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Xd, Yd = torch.from_numpy(X).cuda().repeat(8, 1, 1)[:M].contiguous(), torch.from_numpy(Y).cuda().repeat(8, 1, 1)[:M].contiguous()
lx, ly = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
for _ in range(2):
    pi, pj, pl, c = _hip.fastdtw_l2(Xd, Yd, lx, ly, 1)
t = pi[:, -8:-1].cpu().numpy().astype(np.float64) * 16
print(M, 'pairs; cycles (pyramid, windows, wait-for-costs, first-costs, sweep, backtrace, output):', t.mean(0).astype(int).tolist(), 'total', int(t.mean(0).sum()))
