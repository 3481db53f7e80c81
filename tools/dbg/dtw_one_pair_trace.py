import numpy as np, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip
rng = np.random.RandomState(0)
X = np.zeros((1, 900, 25)); Y = np.zeros((1, 900, 25))
X[0, :812] = np.cumsum(rng.randn(812, 25), 0) * 0.1
Y[0, :777] = np.cumsum(rng.randn(777, 25), 0) * 0.1
for _ in range(6):
    _hip.fastdtw_host(X, Y)
import torch
Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
lx = torch.tensor([812], dtype=torch.int32, device="cuda"); ly = torch.tensor([777], dtype=torch.int32, device="cuda")
for _ in range(5):
    _hip.fastdtw_l2(Xd, Yd, lx, ly)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _hip.fastdtw_l2(Xd, Yd, lx, ly); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print("fastdtw kernel(s) for ONE pair, device-resident, HIP events: median %.1f us (min %.1f)" % (np.median(ts) * 1e3, min(ts) * 1e3))
