"""Kernel time of ONE utterance on every kernel that takes it (device-resident, HIP events, median of 200): what the literal
paramgen.mlpg call waits for between its transfers.  Shapes: BASELINE config 1 (T = 100, 2 dims), one config-2 utterance
(T = 1000, 60 dims), T = 2000 x 60, float64 and float32."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
names = {0: "auto", 1: "generic", 2: "wave", 3: "strip"}


def t_of(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


for dt in (torch.float64, torch.float32):
    for B, T, sd in ((1, 100, 2), (1, 1000, 60), (1, 2000, 60), (1, 500, 60), (4, 1000, 60)):
        m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
        v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
        row = []
        for algo in (0, 2, 3, 1):
            try:
                row.append("%s %.1f us" % (names[algo], t_of(lambda: _hip.forward(m, v, W, algo=algo, want_status=True))))
            except Exception as e:  # noqa: BLE001
                row.append("%s n/a" % names[algo])
        print("%s B=%d T=%d sd=%d: %s" % (str(dt).split(".")[-1], B, T, sd, " | ".join(row)), flush=True)
