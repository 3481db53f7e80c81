"""Literal host-memory calls from several Python threads at once (ctypes releases the GIL around the C call; the library serialises
host calls on one mutex and shares its staging buffers, helper threads and its table of recently seen arrays between them): each
thread draws random problems -- paramgen.mlpg / mlpg_batch / mlpg_grad, now and then DTWAligner.transform on a pair -- and checks
every result against the C oracle (computed in the same thread).  usage: python tools/dbg/lit_threads_soak.py [seconds] [threads] [seed]"""
import os
import sys
import threading
import time

import numpy as np
import torch  # noqa: F401  (imported here, before the threads start: the aligner imports it lazily)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402
from nnmnkwii_amd.preprocessing.alignment import DTWAligner  # noqa: E402
from oracle import dtw as OD  # noqa: E402
from oracle import mlpg as O  # noqa: E402

O.build()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
W2 = W3[:2]
errors = []
counts = [0] * nthreads
t_end = time.time() + seconds


def worker(k):
    rng = np.random.RandomState(seed * 1000 + k)
    keep = []          # a few arrays handed in again later (the direct-copy rule looks at addresses)
    try:
        while time.time() < t_end and not errors:
            w = W3 if rng.rand() < 0.7 else W2
            nw = len(w)
            kind = rng.randint(10)
            if kind == 9:
                a, b = int(rng.randint(20, 300)), int(rng.randint(20, 300))
                X = np.zeros((1, 320, 8))
                Y = np.zeros((1, 320, 8))
                X[0, :a] = np.cumsum(rng.randn(a, 8), 0)
                Y[0, :b] = np.cumsum(rng.randn(b, 8), 0)
                Xa, Ya = DTWAligner().transform((X, Y))
                Xo, Yo = OD.dtw_align(X, Y)[:2]
                assert np.array_equal(Xa, Xo) and np.array_equal(Ya, Yo), "thread %d: DTW mismatch" % k
            else:
                if keep and rng.rand() < 0.3:
                    m, v = keep[rng.randint(len(keep))]
                else:
                    T = int(rng.choice([1, 3, 100, 700, 1000, 2000, 3000]))
                    sd = int(rng.choice([1, 2, 25, 60]))
                    dt = np.float64 if rng.rand() < 0.7 else np.float32
                    m = rng.randn(T, nw * sd).astype(dt)
                    v = (rng.rand(T, nw * sd) + 0.1).astype(dt)
                    if len(keep) < 6:
                        keep.append((m, v))
                    else:
                        keep[rng.randint(6)] = (m, v)
                if m.shape[1] % nw:
                    continue
                T, sd, dt = m.shape[0], m.shape[1] // nw, m.dtype
                tol = 1e-9 if dt == np.float64 else 5e-5
                if kind < 6:
                    y = G.mlpg(m, v, w)
                    yo = O.mlpg(m, v, w)
                    sc = np.abs(yo).max(axis=0) + 1e-300
                    assert float((np.abs(y.astype(np.float64) - yo) / sc).max()) <= tol, "thread %d: mlpg T=%d sd=%d %s" % (k, T, sd, dt)
                else:
                    go = rng.randn(T, sd).astype(dt)
                    g = G.mlpg_grad(m, v, w, go)
                    g2 = G.mlpg_grad(m, v, w, go)
                    assert np.array_equal(g, g2), "thread %d: mlpg_grad not reproducible T=%d sd=%d" % (k, T, sd)
                    if T <= 100:
                        gr = O.mlpg_grad(m, v.astype(np.float64), w, go).astype(np.float64)
                        # (against the whole gradient's maximum: at T = 3 a delta column is a difference of two nearly equal numbers,
                        # 1e-5 of the gradient's size, and the float32 inputs' rounding shows in it -- tools/dbg/grad_small_T.py)
                        assert float(np.abs(g.astype(np.float64) - gr).max()) <= 5e-5 * (np.abs(gr).max() + 1e-300), "thread %d: mlpg_grad T=%d sd=%d" % (k, T, sd)
            counts[k] += 1
    except Exception as e:  # noqa: BLE001
        errors.append("%s: %s" % (type(e).__name__, e))


ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
for t in ths:
    t.start()
for t in ths:
    t.join()
L = _hip.lib()
print("threads soak: %d threads, %.0f s, calls per thread %s, short-path calls %d copied + %d direct; %s"
      % (nthreads, seconds, counts, L.mlpg_hip_launch_count(10), L.mlpg_hip_launch_count(11), ("FAILED: " + errors[0]) if errors else "no mismatch"))
sys.exit(1 if errors else 0)
