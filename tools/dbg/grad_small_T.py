"""paramgen.mlpg_grad against the oracle's dense gradient at very short utterances (T = 1 .. 8), every width / dtype / window set,
single-threaded: where does the relative deviation (of the column maximum) come from?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import paramgen as G  # noqa: E402
from oracle import mlpg as O  # noqa: E402

O.build()
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
W2 = W3[:2]
rng = np.random.RandomState(5)
worst = {}
for it in range(300):
    for T in (1, 2, 3, 4, 5, 8):
        for sd in (1, 2, 25, 60):
            for w in (W3, W2):
                for dt in (np.float64, np.float32):
                    nw = len(w)
                    m = rng.randn(T, nw * sd).astype(dt)
                    v = (rng.rand(T, nw * sd) + 0.1).astype(dt)
                    go = rng.randn(T, sd).astype(dt)
                    g = G.mlpg_grad(m, v, w, go)
                    gr = O.mlpg_grad(m, v.astype(np.float64), w, go).astype(np.float64)
                    sc = np.abs(gr).max(axis=0) + 1e-300
                    e = float((np.abs(g.astype(np.float64) - gr) / sc).max())
                    key = (T, sd, nw, dt.__name__)
                    if e > worst.get(key, (0,))[0]:
                        col = int(np.argmax((np.abs(g.astype(np.float64) - gr) / sc).max(axis=0)))
                        worst[key] = (e, float(sc[col]), float(np.abs(gr).max()))
for key in sorted(worst, key=lambda k: -worst[k][0])[:12]:
    print(key, "worst rel err %.3e (that column's max |grad| %.3e, the whole gradient's %.3e)" % worst[key])
