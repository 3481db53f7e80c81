"""Wall clock of paramgen.mlpg_batch on numpy arrays at the config-2 shape (256 x 1000 x 180 float64, per-frame
variances): into a fresh output array every call (what a user sees), pageable and pinned inputs."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from nnmnkwii_amd import paramgen as G

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
B, T, sd = 256, 1000, 60
M_ = rng.randn(B, T, 3 * sd)
V_ = rng.rand(B, T, 3 * sd) + 0.1
for name, dev in (("one device", None), ("device listed twice", [0, 0])):
    for rep in range(3):
        G.mlpg_batch(M_, V_, W3, device=dev)
    ts = []
    keep = []
    for rep in range(8):
        t0 = time.perf_counter()
        y = G.mlpg_batch(M_, V_, W3, device=dev)
        ts.append(time.perf_counter() - t0)
        keep.append(y)              # every call gets FRESH output pages
    print("%-22s pageable in, fresh out: median %.2f ms  min %.2f ms" % (name, 1e3 * np.median(ts), 1e3 * np.min(ts)))
    del keep
Mp, Vp = _hip.pinned_empty(M_.shape), _hip.pinned_empty(V_.shape)
Mp[...] = M_
Vp[...] = V_
ts = []
keep = []
for rep in range(8):
    t0 = time.perf_counter()
    y = G.mlpg_batch(Mp, Vp, W3)
    ts.append(time.perf_counter() - t0)
    keep.append(y)
print("%-22s pinned in,   fresh out: median %.2f ms  min %.2f ms" % ("one device", 1e3 * np.median(ts), 1e3 * np.min(ts)))
