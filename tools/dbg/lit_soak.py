"""Random soak of the short host path of mlpg_hip_forward_host (one small numpy -> numpy call: paramgen.mlpg / mlpg_batch with at
most 6 MB of input, now and then 24 MB: the arrays the runtime copies directly) against the C oracle: random T (1 .. 2500), static dims (1 .. 70), batch sizes (1 .. 6), window sets (static only,
two and three windows, the reference's 5-tap windows, asymmetric, scaled dynamic windows), per-frame / global / unit variances of
log-normal spread 0 / 1 / 3, float64 / float32, ragged lengths with junk in the padding, occasional negative variances (the reference's
LinAlgError with its k), non-contiguous inputs.  Every AUTO route is reached through pinned-memory outputs (and, below 48 KB, inputs).
usage: python tools/dbg/lit_soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402
from oracle import mlpg as O  # noqa: E402

O.build()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.RandomState(seed)
WSETS = {
    "static": [(0, 0, np.array([1.0]))],
    "std2": [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5]))],
    "std3": [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))],
    "wide3": [(0, 0, np.array([1.0])), (2, 2, np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0), (2, 2, np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0)],
    "asym2": [(0, 0, np.array([1.0])), (1, 0, np.array([-1.0, 1.0]))],
    "dyn4": [(0, 0, np.array([1.0])), (1, 1, 4.0 * np.array([-0.5, 0.0, 0.5])), (1, 1, 4.0 * np.array([1.0, -2.0, 1.0]))],
}
L = _hip.lib()
t_end = time.time() + seconds
n = n_fail = n_direct = n_copied = n_big = n_bwd = 0
worst = {np.float64: 0.0, np.float32: 0.0}
routes0 = [int(L.mlpg_hip_launch_count(k)) for k in range(12)]
while time.time() < t_end:
    wname = list(WSETS)[rng.randint(len(WSETS))]
    w = WSETS[wname]
    nw = len(w)
    dt = np.float64 if rng.rand() < 0.6 else np.float32
    B = 1 if rng.rand() < 0.6 else int(rng.randint(2, 7))
    T = int(rng.choice([1, 2, 3, 5, 17, 64, 100, 333, 1000, 2049])) if rng.rand() < 0.4 else int(rng.randint(1, 2500))
    sd = int(rng.choice([1, 2, 5, 25, 60, 64, 70])) if rng.rand() < 0.6 else int(rng.randint(1, 71))
    cap = (24 << 20) if rng.rand() < 0.15 else (6 << 20)      # (arrays of 1.2 MB and more go straight from / to the caller's memory)
    if cap > (6 << 20) and B > 1:
        T = int(rng.randint(1500, 4000))
    while B * T * nw * sd * np.dtype(dt).itemsize * 2 > cap and T > 1:
        T //= 2
    spread = [0.0, 1.0, 3.0][rng.randint(3)]
    M = rng.randn(B, T, nw * sd).astype(dt)
    V = np.exp(spread * rng.randn(B, T, nw * sd)).astype(dt) * (rng.rand(B, T, nw * sd).astype(dt) + 0.1)
    mode = rng.randint(3)
    var = [V, V[0, 0].copy(), None][mode]
    lengths = None
    if B > 1 and rng.rand() < 0.5:
        lengths = rng.randint(0, T + 1, size=B).astype(np.int32)
        lengths[rng.randint(B)] = T
        for b in range(B):
            M[b, lengths[b]:] = 1e30
    bad = mode == 0 and T >= 3 and rng.rand() < 0.08
    if bad:
        b_ = rng.randint(B)
        tl = T if lengths is None else int(lengths[b_])
        if tl >= 3:
            V[b_, rng.randint(tl), rng.randint(sd)] *= -1e-9                  # a static-window variance: its precision swamps the row, the pivot fails
        else:
            bad = False
    r0 = (int(L.mlpg_hip_launch_count(10)), int(L.mlpg_hip_launch_count(11)))
    vo = var if var is not None else np.ones(nw * sd, dtype=dt)
    yo, sto, rc = O.mlpg_batch(M, vo, w, lengths)
    try:
        if B == 1 and lengths is None and mode != 2:
            if rng.rand() < 0.2:                 # a non-contiguous view of the means
                Mp = np.zeros((T, 2 * nw * sd), dtype=dt)
                Mp[:, ::2] = M[0]
                y = G.mlpg(Mp[:, ::2], var[0] if mode == 0 else var, w)[None]
            else:
                y = G.mlpg(M[0], var[0] if mode == 0 else var, w)[None]
        else:
            y = G.mlpg_batch(M, var, w, lengths)
        raised = None
    except np.linalg.LinAlgError as e:
        raised = str(e)
    r1 = (int(L.mlpg_hip_launch_count(10)), int(L.mlpg_hip_launch_count(11)))
    n_copied += r1[0] - r0[0]
    n_direct += r1[1] - r0[1]
    n_big += (r1[0] - r0[0]) + (r1[1] - r0[1]) == 0
    n += 1
    first_bad = sto.ravel()[np.flatnonzero(sto.ravel())[0]] if sto.any() else 0
    if first_bad:
        n_fail += 1
        want = "%d-th leading minor not positive definite" % first_bad
        assert raised == want, ("case %d: %s T=%d sd=%d B=%d mode %d %s: raised %r, the oracle says %r" % (n, wname, T, sd, B, mode, dt.__name__, raised, want))
        continue
    assert raised is None, ("case %d raised %r but the oracle solved it" % (n, raised))
    if bad:
        continue      # (a negative variance that did not break a pivot: an indefinite system both sides "solved"; values not compared)
    assert y.dtype == dt and y.shape == yo.shape
    for b in range(B):
        tl = T if lengths is None else int(lengths[b])
        if tl == 0:
            continue
        scale = np.abs(yo[b, :tl]).max(axis=0) + 1e-300
        err = float((np.abs(y[b, :tl].astype(np.float64) - yo[b, :tl]) / scale).max())
        worst[dt] = max(worst[dt], err)
        # (wide log-normal variances: the systems' condition numbers reach 1e8; the bound is that of the other soaks)
        tol = (1e-7 if spread >= 3.0 else 1e-9) if dt == np.float64 else 2e-3 if spread >= 3.0 else 5e-5
        assert err <= tol, ("case %d: %s T=%d sd=%d B=%d mode %d %s spread %g: rel err %.3e" % (n, wname, T, sd, B, mode, dt.__name__, spread, err))
        assert not y[b, tl:].any()
    # the literal backward call on the same variances (mlpg_hip_backward_host) against mlpg_hip_backward on device copies: same
    # kernels and routing, so the two must agree bit for bit; small ones also against the oracle's dense mlpg_grad
    if mode != 2 and not bad and n % 3 == 0:
        import torch
        go = rng.randn(B, T, sd).astype(dt)
        od = np.float32 if rng.rand() < 0.7 else np.float64
        g, st = _hip.backward_host(var if mode == 1 else V, go, w, nw * sd, out_dtype=od, lengths=lengths)
        vd = torch.from_numpy(var if mode == 1 else V).cuda()
        gd, std = _hip.backward(vd, torch.from_numpy(go).cuda(), w, nw * sd, lengths=None if lengths is None else torch.from_numpy(lengths).cuda(),
                                out_dtype=torch.float32 if od == np.float32 else torch.float64)
        assert np.array_equal(st.ravel(), std.cpu().numpy().ravel()) and np.array_equal(g, gd.cpu().numpy()), (
            "case %d backward: %s T=%d sd=%d B=%d mode %d %s" % (n, wname, T, sd, B, mode, dt.__name__))
        n_bwd += 1
        if T <= 200 and lengths is None and dt == np.float64 and spread < 3.0:
            gr = O.mlpg_grad(np.zeros((T, nw * sd)), V[0] if mode == 0 else np.tile(var, (T, 1)), w, go[0]).astype(np.float64)
            sc = np.abs(gr).max(axis=0) + 1e-300
            assert float((np.abs(g[0].astype(np.float64) - gr) / sc).max()) <= 2e-6, "case %d backward vs oracle" % n
routes1 = [int(L.mlpg_hip_launch_count(k)) for k in range(12)]
names = "generic wave strip strip-multi const fused chunk fir const-multi strip-tr short-copied short-direct".split()
print("lit soak: %d calls in %.0f s (%d of them with a failing pivot: the reference's exception and k), %d on the short path with copied inputs, "
      "%d with the kernel reading pinned inputs, %d beside it; %d backward calls (mlpg_hip_backward_host == mlpg_hip_backward bit for bit); "
      "worst rel err float64 %.2e float32 %.2e; no mismatch"
      % (n, seconds, n_fail, n_copied, n_direct, n_big, n_bwd, worst[np.float64], worst[np.float32]))
print("kernel launches by family:", {nm: b - a for nm, a, b in zip(names, routes0, routes1) if b != a})
