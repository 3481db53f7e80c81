"""Where DTWAligner.transform spends its time for one GPU's share of config 4 (128 pairs, T in [700, 900], 25 dims, float64)."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from nnmnkwii_amd.preprocessing.alignment import DTWAligner

rng = np.random.RandomState(1234)
N, Tm, D = 128, 900, 25
X = np.zeros((N, Tm, D))
Y = np.zeros((N, Tm, D))
for n in range(N):
    tx, ty = rng.randint(700, 901, 2)
    X[n, :tx] = np.cumsum(rng.randn(tx, D), 0) * 0.1
    Y[n, :ty] = np.cumsum(rng.randn(ty, D), 0) * 0.1
al = DTWAligner()
for _ in range(3):
    al.transform((X, Y))


def t(fn, n=10):
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r


ms, _ = t(lambda: al.transform((X, Y)))
print("transform (device-tensor route)      %.2f ms" % ms)
dev = torch.device("cuda", 0)
ms, (Xd, Yd) = t(lambda: (torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)))
print("  upload X, Y (torch, pageable)      %.2f ms  (%.1f MB)" % (ms, (X.nbytes + Y.nbytes) / 1e6))
ms, (lx, ly) = t(lambda: (_hip.trim_lengths(Xd), _hip.trim_lengths(Yd)))
print("  trim                               %.2f ms" % ms)
ms, (pi, pj, pl, cost) = t(lambda: _hip.fastdtw_l2(Xd, Yd, lx, ly, 1))
print("  fastdtw                            %.2f ms" % ms)
ms, plh = t(lambda: pl.cpu().numpy())
print("  path lengths to the host           %.2f ms" % ms)
T_out = max(Tm, int(plh.max()))
ms, (ga, gb) = t(lambda: (_hip.gather_path(Xd, pi, pl, T_out), _hip.gather_path(Yd, pj, pl, T_out)))
print("  gather                             %.2f ms" % ms)
ms, _ = t(lambda: (ga.cpu().numpy(), gb.cpu().numpy()))
print("  download aligned arrays            %.2f ms  (%.1f MB)" % (ms, 2 * ga.numel() * 8 / 1e6))
al2 = DTWAligner()
al2._HOST_ENTRY_BYTES = 0
for _ in range(2):
    al2.transform((X, Y))
ms, _ = t(lambda: al2.transform((X, Y)))
print("transform (host-entry route)         %.2f ms" % ms)
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    al.transform((X, Y))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
