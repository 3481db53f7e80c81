"""Constant-coefficient kernel: time against T (B sequences of 60 dims, float64, global variances) -- slope = cost of
a 128-frame super-step under full load, intercept = what a launch costs before and after its super-steps."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def t_of(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):         # queued ahead: the host's launch latency must not sit between the events
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


for B in (int(a) for a in (sys.argv[1:] or ["256", "512", "64"])):
    for T in (16, 128, 256, 512, 1024, 2048):
        sd = 60
        m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda")
        vg = torch.rand(3 * sd, dtype=torch.float64, device="cuda") + 0.1
        med, mn = t_of(lambda: _hip.forward(m, vg, W3, None, algo=5, want_status=False))
        by = 32.0 * sd * B * T
        print("B %4d T %5d: median %.4f ms  min %.4f  (%.0f GB/s, frac %.3f)" % (B, T, med, mn, by / med / 1e6, by / med / 1e6 / 8000))
