"""Random soak of the FIR form of unit-variance MLPG (algo 7, float32, no lengths) and of the training step in that form:
forward against the C oracle, backward against the natural-order kernel, the step's loss / gradient against those two.
usage: python tools/dbg/fir_soak.py [seconds [seed]]       (written at the end of round 4; first run is due in round 5)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cases import WINDOW_SETS  # noqa: E402
from oracle import mlpg as O  # noqa: E402


def soak(budget, seed):
    import torch
    from nnmnkwii_amd import _hip
    rng = np.random.RandomState(seed)
    O.build()
    t0 = time.time()
    n, bad = 0, None
    tol = 5e-6
    while time.time() - t0 < budget and bad is None:
        wname = ["std3", "std2", "asym2", "wide3"][rng.randint(4)]
        win = WINDOW_SETS[wname]
        nw = len(win)
        B = int(rng.choice([1, 2, 3, 7, 64, 70]))
        T = int(rng.choice([96, 97, 119, 120, 121, 128, 160, 255, 256, 257, 500, 777, 1000, 1025, 2100]))
        sd = int(rng.choice([1, 2, 5, 25, 60, 64, 65, 80, 130]))
        if T * sd * B > 2.5e6:
            B = max(1, int(2.5e6 / (T * sd)))
        m = rng.randn(B, T, nw * sd).astype(np.float32)
        ref, st, rc = O.mlpg_batch(m, np.ones(nw * sd, dtype=np.float32), win)
        assert rc == 0
        md = torch.from_numpy(m).cuda()
        y, status = _hip.forward(md, None, win, None, algo=_hip.ALGO_FIR)
        scale = np.abs(ref).reshape(-1, sd).max(axis=0)
        scale = np.where(scale == 0, 1.0, scale)
        err = float((np.abs(y.cpu().numpy().astype(np.float64) - ref).reshape(-1, sd) / scale).max())
        if not err <= tol or int(status.abs().sum()) != 0:
            bad = ("forward", wname, B, T, sd, err)
            break
        g = torch.from_numpy(rng.randn(B, T, sd).astype(np.float32)).cuda()
        gref, _ = _hip.backward(None, g, win, nw * sd, out_dtype=torch.float32, algo=_hip.ALGO_GENERIC)
        gf, _ = _hip.backward(None, g, win, nw * sd, out_dtype=torch.float32, algo=_hip.ALGO_FIR)
        err = float((gf - gref).abs().max()) / max(1e-30, float(gref.abs().max()))
        if not err <= tol:
            bad = ("backward", wname, B, T, sd, err)
            break
        tg = torch.from_numpy(rng.rand(B, T, sd).astype(np.float32)).cuda()
        n0 = _hip.lib().mlpg_hip_launch_count(7)
        loss, grad, ys, _ = _hip.unit_mse_step(md, tg, win, want_y=True)
        if _hip.lib().mlpg_hip_launch_count(7) != n0 + 2:
            bad = ("step not in the FIR form", wname, B, T, sd)
            break
        e = ys.double() - tg.double()
        lref = float((e * e).mean())
        gsref, _ = _hip.backward(None, (2.0 * e / e.numel()).float(), win, nw * sd, out_dtype=torch.float32, algo=_hip.ALGO_GENERIC)
        e1 = float((ys - y).abs().max()) / max(1e-30, float(y.abs().max()))
        e2 = abs(float(loss) - lref) / max(1e-30, lref)
        e3 = float((grad - gsref).abs().max()) / max(1e-30, float(gsref.abs().max()))
        if not (e1 <= 1e-6 and e2 <= 1e-6 and e3 <= 2 * tol):
            bad = ("step", wname, B, T, sd, e1, e2, e3)
            break
        n += 1
    return n, bad


if __name__ == "__main__":
    r = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0, int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    print("cases", r[0], "mismatch", r[1])
