"""Random soak of every MLPG kernel (natural-order, wave-per-system, strip, constant-coefficient, chunked) against the C oracle (forward)
and against each other (backward: the natural-order kernel is the reference, itself pinned by tests/): random batch sizes,
lengths (ragged), static dims 1..130, the three variance modes, float32 / float64, window sets of extent <= 1 (all kernels) and
the 5-tap set (natural-order and chunked kernels vs the oracle).   usage: python tools/dbg/mlpg_algos_soak.py [seconds [seed]]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
from cases import WINDOW_SETS  # noqa: E402
from nnmnkwii_amd import _hip  # noqa: E402
from oracle import mlpg as O  # noqa: E402


def soak(budget=40.0, seed=99):
    rng = np.random.RandomState(seed)
    O.build()
    t0 = time.time()
    n = 0
    bad = None
    names = {1: "generic", 2: "wave", 3: "strip", 5: "const", 6: "chunk"}
    while time.time() - t0 < budget and bad is None:
        dt = [np.float64, np.float32][rng.randint(2)]
        tol = 2e-9 if dt == np.float64 else 3e-6
        wname = ["std3", "std2", "static", "wide3", "zero2", "asym2"][rng.randint(6)]
        win = WINDOW_SETS[wname]
        nw = len(win)
        B = int(rng.randint(1, 7))
        T = int(rng.choice([1, 2, 3, 17, 64, 65, 130, 500, 1000, 1025, 2049, 2500]))
        sd = int(rng.choice([1, 2, 5, 25, 60, 64, 65, 80, 130]))
        if T * sd * B > 1.2e6:
            sd = max(1, int(1.2e6 / (T * B)))
        vmode = rng.randint(3)
        m = rng.randn(B, T, nw * sd).astype(dt)
        lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
        lengths[rng.randint(B)] = T
        for b in range(B):
            m[b, lengths[b]:] = 0
        spread = [0.0, 1.0, 3.0][rng.randint(3)]          # log-normal variances: the wider, the longer the coupling
        if vmode == 0:
            v = np.exp(spread * rng.randn(B, T, nw * sd)).astype(dt) * 0.5 + 0.05
        elif vmode == 1:
            v = (np.exp(spread * rng.randn(nw * sd)) * 0.5 + 0.05).astype(dt)
        else:
            v = None
        vo = v if v is not None else np.ones(nw * sd, dtype=dt)
        ref, st, rc = O.mlpg_batch(m, vo, win, lengths)
        if rc != 0:
            continue
        md = torch.from_numpy(m).cuda()
        vd = None if v is None else torch.from_numpy(v).cuda()
        Ld = torch.from_numpy(lengths).cuda()
        scale = max(1.0, float(np.abs(ref).max()))
        slack = 1.0 if spread < 3.0 else 1e4              # ill-conditioned systems: the kernels agree with each other to ~1e-6
        for algo in (1, 2, 3, 5, 6):
            try:
                out, status = _hip.forward(md, vd, win, Ld, algo=algo)
            except _hip.HipExtensionError:
                continue                                    # this kernel does not take this problem (extent > 1, T too long)
            err = float(np.abs(out.cpu().numpy() - ref).max())
            if not (err <= tol * slack * scale) or int(status.abs().sum()) != 0:
                bad = ("forward", names[algo], dt.__name__, wname, B, T, sd, vmode, spread, err, scale)
                break
        if bad is None:
            g = torch.from_numpy(rng.randn(B, T, sd).astype(dt)).cuda()
            gref, _ = _hip.backward(vd, g, win, nw * sd, lengths=Ld, out_dtype=md.dtype, algo=1)
            gs = max(1e-30, float(gref.abs().max()))
            for algo in (2, 3, 5, 6):
                try:
                    go, status = _hip.backward(vd, g, win, nw * sd, lengths=Ld, out_dtype=md.dtype, algo=algo)
                except _hip.HipExtensionError:
                    continue
                err = float((go - gref).abs().max())
                if not (err <= tol * slack * 10 * gs):
                    bad = ("backward", names[algo], dt.__name__, wname, B, T, sd, vmode, spread, err, gs)
                    break
        n += 1
    return n, bad


if __name__ == "__main__":
    r = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0, int(sys.argv[2]) if len(sys.argv) > 2 else 99)
    print("cases", r[0], "mismatch", r[1])
