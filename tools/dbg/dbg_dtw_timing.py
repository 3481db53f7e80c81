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
t = pi[:, -12:-1].cpu().numpy().astype(np.float64) * 16
m = t.mean(0)
print(M, 'pairs; cycles (pyramid, windows, wait-for-costs, first-costs, sweep, backtrace, output):',
      [int(m[0]), int(m[1]), int(m[2]), int(m[3]), int(m[4]), int(m[5] + m[7] + m[8] + m[9]), int(m[6])], 'total', int(m.sum()),
      '| backtrace = candidate offsets %d + pass 1 %d + doubling/select/prefix %d + pass 2 %d' % (m[7], m[8], m[9], m[5]))
