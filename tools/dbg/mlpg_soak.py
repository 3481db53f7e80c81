"""Random soak of the round-3 MLPG paths against their slower equivalents (no oracle needed: both sides are the HIP library;
the slow sides are pinned against the oracle by tests/):
  * mlpg_hip_forward_streams (streams merged into one strip launch, stream pieces) vs mlpg_hip_forward on a dense copy of
    every stream's columns;
  * mlpg_hip_unit_mse_step (the backward solve as a second right-hand side) vs forward + MSE gradient + backward launches.
usage: python tools/dbg/mlpg_soak.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
from cases import WINDOW_SETS  # noqa: E402
from nnmnkwii_amd import _hip  # noqa: E402

STD3 = WINDOW_SETS["std3"]
def soak(budget=40.0, seed=4711):
    rng = np.random.RandomState(seed)
    t0 = time.time()
    n_streams = n_fused = n_nolen = 0
    tr0 = int(_hip.lib().mlpg_hip_launch_count(9))
    bad = None
    while time.time() - t0 < budget and bad is None:
        dt = [np.float64, np.float32][rng.randint(2)]
        tol = 1e-9 if dt == np.float64 else 3e-6
        if rng.rand() < 0.6:
            # ---- multi-stream ----
            B = int(rng.randint(1, 9)) if rng.rand() < 0.6 else int(rng.randint(9, 140))   # (enough utterances for the transposed form's lane groups)
            T = int(rng.choice([40, 300, 700, 1100, 1500, 2100]))
            if B > 40:
                T = min(T, 1100)
            k = int(rng.randint(2, 6))
            sds = [int(rng.choice([1, 2, 3, 5, 7, 20, 40, 60, 64, 66])) for _ in range(k)]
            passthru = [rng.rand() < 0.2 for _ in range(k)]
            cols, c = [], 0
            for sd, pt in zip(sds, passthru):
                cols.append(c)
                c += sd if pt else 3 * sd
                c += int(rng.randint(0, 3))           # unused columns between streams
            D = c
            m = rng.randn(B, T, D).astype(dt)
            v = (rng.rand(B, T, D) + 0.1).astype(dt)
            lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
            lengths[rng.randint(B)] = T
            no_lengths = rng.rand() < 0.4     # also batches without a lengths vector (round 5's first form of the transposed strip kernel needed that; now either way)
            if no_lengths:
                lengths[:] = T
            for b in range(B):
                m[b, lengths[b]:] = 0
            md, vd, Ld = torch.from_numpy(m).cuda(), torch.from_numpy(v).cuda(), (None if no_lengths else torch.from_numpy(lengths).cuda())
            n_nolen += int(no_lengths)
            streams = [(cols[i], sds[i], None if passthru[i] else STD3) for i in range(k)]
            # variance mode: per frame (merged strip launch), global (D,) or unit (round 5: merged constant-coefficient launch)
            vmode = ["frame", "global", "unit"][rng.randint(3)]
            if vmode == "frame":
                algo = _hip.ALGO_STRIP if rng.rand() < 0.5 else _hip.ALGO_AUTO
                var_all = vd
            else:
                algo = _hip.ALGO_CONST if rng.rand() < 0.7 else _hip.ALGO_AUTO
                var_all = torch.from_numpy((rng.rand(D) + 0.1).astype(dt)).cuda() if vmode == "global" else None
            out, st = _hip.forward_streams(md, var_all, streams, Ld, algo=algo)
            o0 = 0
            for (c0, sd, win) in streams:
                if win is not None:
                    var_s = None if var_all is None else (var_all[:, :, c0:c0 + 3 * sd] if vmode == "frame" else var_all[c0:c0 + 3 * sd]).contiguous()
                    dense, dst = _hip.forward(md[:, :, c0:c0 + 3 * sd].contiguous(), var_s, STD3, Ld, algo=_hip.ALGO_WAVE if T <= 2048 else _hip.ALGO_GENERIC)
                    err = float((out[:, :, o0:o0 + sd] - dense).abs().max())
                    tol_s = tol if (vmode == "frame" or dt == np.float64) else 6e-6   # (different kernels on both sides in float32)
                    if not (err <= tol_s * max(1.0, float(dense.abs().max()))) or int(st[:, o0:o0 + sd].abs().sum()) != 0:
                        bad = ("streams", dt.__name__, vmode, B, T, streams, algo, c0, sd, err)
                        break
                o0 += sd
            n_streams += 1
        else:
            # ---- fused unit-variance step ----
            B = int(rng.randint(1, 40))
            T = int(rng.choice([1, 2, 5, 37, 64, 200, 256, 257, 500, 512, 513, 900]))
            sd = int(rng.randint(1, 70))
            m = torch.from_numpy(rng.rand(B, T, 3 * sd).astype(dt)).cuda()
            tg = torch.from_numpy(rng.rand(B, T, sd).astype(dt)).cuda()
            lengths = None
            if rng.rand() < 0.5:
                ln = rng.randint(1, T + 1, size=B).astype(np.int32)
                lengths = torch.from_numpy(ln).cuda()
            loss, grad, y, _ = _hip.unit_mse_step(m, tg, STD3, lengths=lengths, want_y=True)
            yr, _ = _hip.forward(m, None, STD3, lengths)
            mask = torch.ones(B, T, 1, device="cuda", dtype=m.dtype)
            if lengths is not None:
                mask = (torch.arange(T, device="cuda")[None, :, None] < lengths[:, None, None]).to(m.dtype)
            g = 2.0 * (yr - tg) * mask / float(B * T * sd)
            gr, _ = _hip.backward(None, g.contiguous(), STD3, 3 * sd, lengths=lengths, out_dtype=m.dtype)
            lr = float((((yr - tg) * mask) ** 2).sum() / float(B * T * sd))
            e1 = float((y * mask - yr * mask).abs().max())
            e2 = float((grad - gr).abs().max())
            if not (e1 <= tol and e2 <= tol * max(1e-6, float(gr.abs().max())) + 1e-30 and abs(float(loss) - lr) <= 10 * tol * max(lr, 1e-30)):
                bad = ("fused", dt.__name__, B, T, sd, lengths is not None, e1, e2, float(loss), lr)
            n_fused += 1
    return n_streams, n_fused, bad, n_nolen, int(_hip.lib().mlpg_hip_launch_count(9)) - tr0


if __name__ == "__main__":
    r = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 40.0, int(sys.argv[2]) if len(sys.argv) > 2 else 4711)
    print("multi-stream cases", r[0], "(%d without lengths; %d launches of the transposed strip form)" % (r[3], r[4]), "fused cases", r[1], "mismatch", r[2])
