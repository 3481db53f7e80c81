import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from nnmnkwii_amd import _hip
from oracle import dtw as OD
rng = np.random.RandomState(1)
def run(pairs, radius, Tpad=None):
    N = len(pairs); D = pairs[0][0].shape[1]
    Tx = max(len(x) for x, _ in pairs); Ty = max(len(y) for _, y in pairs)
    if Tpad: Tx, Ty = max(Tx, Tpad[0]), max(Ty, Tpad[1])
    X, Y = np.zeros((N, Tx, D)), np.zeros((N, Ty, D))
    for n, (x, y) in enumerate(pairs):
        X[n, :len(x)] = x; Y[n, :len(y)] = y
    lx = torch.tensor([len(x) for x, _ in pairs], dtype=torch.int32, device="cuda")
    ly = torch.tensor([len(y) for _, y in pairs], dtype=torch.int32, device="cuda")
    pi, pj, pl, cost = _hip.fastdtw_l2(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), lx, ly, radius)
    return pl.cpu().numpy(), cost.cpu().numpy()
for D in (1, 13):
    for radius in (1, 2, 3):
        for tx, ty in ((324, 1), (1, 324), (63, 1), (64, 1), (65, 1), (100, 1), (200, 2), (324, 2), (324, 3), (30, 1), (9, 1), (8, 1)):
            x, y = rng.randn(tx, D), rng.randn(ty, D)
            for pad in (None, (420, 420)):
                pl, c = run([(x, y)], radius, pad)
                d, path = OD.fastdtw(x, y, radius)
                if pl[0] != len(path):
                    print("FAIL D", D, "r", radius, (tx, ty), "pad", pad, "gpu len", pl[0], "oracle", len(path))
print("done")
