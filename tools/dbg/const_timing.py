"""Phase cycle counters of the constant-coefficient kernel (library built with -DMLPG_CONST_TIMING: the counters
overwrite the head of the status array).  Usage: python tools/dbg/const_timing.py [B T sd f64|f32 global|unit [bwd]]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
NAMES = ["other", "pass1a", "B0", "halo+pass1b", "B1", "prefix+pass2", "B2+B3", "suffix", "parked out", "own out/park", "verdict",
         "-", "super-steps", "parked", "sequences", "-"]
NT = 16


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
    W = 8
    n = len(st) // NT
    q = st[:n * NT].reshape(n, NT).astype(np.float64)
    q[:, :12] *= 16
    q = q[q[:, 12] > 0]
    print("waves with work: %d; cycles per super-step by phase (mean over waves | wave 0 | wave 7):" % len(q))
    wv = np.arange(len(q)) % W
    for k in range(11):
        f = lambda sel: q[sel, k].sum() / max(1.0, q[sel, 12].sum())
        print("  %-14s %9.0f | %9.0f | %9.0f" % (NAMES[k], f(slice(None)), f(wv == 0), f(wv == W - 1)))
    print("  total per super-step %.0f cycles; super-steps per wave %.1f; parked share %.2f; sequences per wave %.1f" % (
        q[:, :11].sum() / q[:, 12].sum(), q[:, 12].mean(), q[:, 13].sum() / q[:, 12].sum(), q[:, 14].mean()))


if __name__ == "__main__":
    main()
