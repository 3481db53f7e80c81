"""Index-level numpy emulation of fir_tiles_shared (mlpg_fir_shared.patch): the group -> tile mapping, which wavefront forms which rows
of the shared right-hand side from which frames, which shared rows a tile reads back, which output rows it owns -- checked against
tools/fir_model.py (the specification of the shipped kernel) on the rows the tiles write.   usage: python tools/experimental/fir_shared/model.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fir_model as FM  # noqa: E402
from cases import WINDOW_SETS  # noqa: E402

H, E, TT, NWV = 24, 24, 32, 8


def tiles_shared(x, windows, T, backward):
    """x: means (T, nw*sd) forward / grad_out (T, sd) backward.  Returns (rows written by the tile role, their values)."""
    nw = len(windows)
    mw = max(max(l, u) for l, u, _ in windows)
    EXT = 1 if mw <= 1 else 2
    sd = x.shape[1] if backward else x.shape[1] // nw
    TAP, ok = FM.build_taps(windows)
    assert ok
    tap = TAP[0]
    NO = TT + 2 * EXT if backward else TT
    NB = NO + 2 * H
    XB = EXT if backward else 0
    NROW = NWV * TT + 2 * H + 2 * XB
    RPW = (NROW + NWV - 1) // NWV
    NFA = RPW if backward else RPW + 2 * EXT
    EW = E + EXT if backward else E
    nt = (T + TT - 1) // TT
    ngt = (nt + NWV - 1) // NWV
    lo = [0] + [mw] * (nw - 1)
    span = [T] + [(T - 2 * mw if (mw != 0 and T > 2 * mw) else 0)] * (nw - 1)
    cw = np.zeros((nw, 2 * EXT + 1))
    for w, (l, u, c) in enumerate(windows):
        for k in range(-l, u + 1):
            cw[w, k + EXT] = c[l + k]
    out = {}
    for grp in range(ngt):
        F0 = grp * NWV * TT
        lb = np.full((RPW * NWV, sd), np.nan)
        for wv in range(NWV):                           # ---- phase A
            fA = F0 - H - XB + wv * RPW
            f_first = fA - (0 if backward else EXT)
            if backward:
                for s in range(RPW):
                    t = f_first + s
                    lb[wv * RPW + s] = x[t] if 0 <= t < T else 0.0
            else:
                bbA = np.zeros((RPW, sd))
                for s in range(NFA):
                    t = f_first + s
                    for w in range(nw):
                        lv = 0 <= t - lo[w] < span[w]
                        m = x[t, w * sd:(w + 1) * sd] if lv else np.zeros(sd)
                        for k in range(-EXT, EXT + 1):
                            ib = s - EXT + k
                            if ib < 0 or ib >= RPW or (w == 0 and k != 0):
                                continue
                            bbA[ib] += cw[w, k + EXT] * m
                    if 0 <= s - 2 * EXT < RPW:
                        lb[wv * RPW + s - 2 * EXT] = bbA[s - 2 * EXT]
        assert not np.isnan(lb[:NROW]).any()            # every shared row a tile may read was written
        for wv in range(NWV):                           # ---- phase B
            tile = grp * NWV + wv
            t0 = tile * TT
            if tile >= nt or t0 + TT <= EW or t0 >= T - EW:
                continue
            rows = lb[wv * TT:wv * TT + NB]               # row i <-> frame t0 - XB - H + i
            assert wv * TT + NB <= NROW
            z = np.stack([tap @ rows[r:r + 2 * H + 1] for r in range(NO)])   # z[r] <-> frame t0 - XB + r
            for r in range(TT):
                t = t0 + r
                if not (EW <= t < T - EW):
                    continue
                if not backward:
                    out[t] = z[r]
                else:
                    g = np.zeros(nw * sd)
                    for w in range(nw):
                        lv = 0 <= t - lo[w] < span[w]
                        if not lv:
                            continue
                        for k in range(-EXT, EXT + 1):
                            if w == 0 and k != 0:
                                continue
                            g[w * sd:(w + 1) * sd] += cw[w, k + EXT] * z[r + EXT + k]
                    out[t] = g
    return out


def check():
    n = 0
    for wname in ("std3", "std2", "asym2", "wide3"):
        win = WINDOW_SETS[wname]
        nw = len(win)
        for T in (96, 97, 128, 255, 256, 257, 289, 500, 513, 1000):
            rng = np.random.RandomState(T)
            m = rng.randn(T, nw * 2)
            ref = FM.forward(m, win, T)
            got = tiles_shared(m, win, T, False)
            mw = max(max(l, u) for l, u, _ in win)
            EXT = 1 if mw <= 1 else 2
            assert sorted(got) == list(range(E, T - E)), (wname, T)
            for t, v in got.items():
                assert np.abs(v - ref[t]).max() <= 1e-12 * np.abs(ref).max(), (wname, T, t)
            go = rng.randn(T, 2)
            refb = FM.backward(go, win, T)
            gotb = tiles_shared(go, win, T, True)
            assert sorted(gotb) == list(range(E + EXT, T - E - EXT)), (wname, T)
            for t, v in gotb.items():
                assert np.abs(v - refb[t]).max() <= 1e-12 * np.abs(refb).max(), (wname, T, t)
            n += 1
    return n


if __name__ == "__main__":
    print("cases", check(), "ok")
