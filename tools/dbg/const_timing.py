"""Phase cycle counters of the constant-coefficient kernel (library built with -DMLPG_CONST_TIMING: the counters
overwrite the head of the status array).  Usage: python tools/dbg/const_timing.py [B T sd f64|f32 global|unit [bwd]]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
NAMES = ["setup", "pass1a+1b", "B0+B1", "pre/pub/look", "B2", "pass2", "B3", "suf/pub/look", "B4", "pass3", "stores", "ticket",
         "items", "edge", "lb", "la", "pub fwd", "flag wait fwd", "records fwd", "pub bwd", "flag wait bwd", "records bwd", "-", "-"]
NT = 24


def main():
    a = sys.argv[1:]
    B, T, sd = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (256, 1000, 60)
    dt = torch.float32 if len(a) >= 4 and a[3] == "f32" else torch.float64
    unit = len(a) >= 5 and a[4] == "unit"
    bwd = len(a) >= 6 and a[5] == "bwd"
    m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
    vg = None if unit else torch.rand(3 * sd, dtype=dt, device="cuda") + 0.1
    go = torch.randn(B, T, sd, dtype=dt, device="cuda")
    import os
    shape = int(os.environ.get('MLPG_CONST_SHAPE', '0' if B * ((T + 127) // 128) >= 1024 else '1'))
    W = {0: 4, 1: 2, 2: 8, 3: 4}[shape]
    for rep in range(3):
        if bwd:
            _, st = _hip.backward(vg, go, W3, 3 * sd, None, out_dtype=dt, algo=5)
        else:
            _, st = _hip.forward(m, vg, W3, None, algo=5)
    torch.cuda.synchronize()
    ts = []
    for rep in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if bwd:
            _hip.backward(vg, go, W3, 3 * sd, None, out_dtype=dt, algo=5, want_status=False)
        else:
            _hip.forward(m, vg, W3, None, algo=5, want_status=False)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("B %d T %d sd %d %s %s %s: median %.4f ms, min %.4f" % (B, T, sd, dt, "unit" if unit else "global", "bwd" if bwd else "fwd", np.median(ts), np.min(ts)))
    st = st.cpu().numpy()
    n = (len(st) // NT)
    q = st[:n * NT].reshape(n, NT).astype(np.float64)
    q[:, :12] *= 16
    q[:, 16:] *= 16
    q = q[q[:, 12] > 0]
    wv = np.arange(len(q)) % W
    print("waves with items: %d; cycles per item by phase (wave 0 | other waves):" % len(q))
    for k in range(12):
        a0 = q[wv == 0, k].sum() / q[wv == 0, 12].sum()
        a1 = q[wv != 0, k].sum() / max(1.0, q[wv != 0, 12].sum())
        print("  %-14s %9.0f | %9.0f" % (NAMES[k], a0, a1))
    for k in range(16, 22):
        print("  %-14s %9.0f |" % (NAMES[k], q[wv == 0, k].sum() / q[wv == 0, 12].sum()))
    tot0 = q[wv == 0, :12].sum() / q[wv == 0, 12].sum()
    print("  total per item %.0f cycles; items per wave %.1f; edge chunk share %.3f; look-back %.2f, look-ahead %.2f steps per item" % (
        tot0, q[:, 12].mean(), q[:, 13].sum() / q[:, 12].sum(), q[wv == 0, 14].sum() / q[wv == 0, 12].sum(), q[wv == 0, 15].sum() / q[wv == 0, 12].sum()))


if __name__ == "__main__":
    main()
