"""Config-2 shape (256 x 1000 x 180, per-frame variances, the 3-tap windows) on the STRIP kernel of whatever library
NNMNKWII_AMD_SO selects: forward / backward times, float64 and float32, and the deviation from the natural-order kernel
(meaningless for -DMLPG_STRIP_ABLATE builds, which skip the inter-workgroup level).
usage: python tools/dbg/strip_variant_time.py [fwd|bwd|all] [f64|f32|both]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


what = sys.argv[1] if len(sys.argv) > 1 else "all"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
tag = os.path.basename(os.environ.get("NNMNKWII_AMD_SO", "default"))
B, T, sd = 256, 1000, 60
gen = torch.Generator(device="cuda").manual_seed(1234)
for dt in (torch.float64, torch.float32):
    if which != "both" and which != ("f64" if dt == torch.float64 else "f32"):
        continue
    m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda", generator=gen)
    v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda", generator=gen) + 0.1
    g = torch.randn(B, T, sd, dtype=dt, device="cuda", generator=gen)
    esz = 8 if dt == torch.float64 else 4
    by = esz * 7 * sd * B * T
    if what in ("fwd", "all"):
        f, fmin = timeit(lambda: _hip.forward(m, v, W3, None, algo=3, want_status=False))
        y, st = _hip.forward(m[:8], v[:8], W3, None, algo=3)
        yr, _ = _hip.forward(m[:8], v[:8], W3, None, algo=1)
        err = float((y - yr).abs().max() / yr.abs().max())
        print("%-28s %s forward  %.4f ms (min %.4f)  frac %.3f  dev vs natural-order %.2e  status %d" % (tag, str(dt)[6:], f, fmin, by / f / 1e6 / 8000, err, int(st.abs().sum())), flush=True)
    if what in ("bwd", "all"):
        b, bmin = timeit(lambda: _hip.backward(v, g, W3, 3 * sd, out_dtype=dt, algo=3, want_status=False))
        gr, st = _hip.backward(v[:8], g[:8], W3, 3 * sd, out_dtype=dt, algo=3)
        grr, _ = _hip.backward(v[:8], g[:8], W3, 3 * sd, out_dtype=dt, algo=1)
        err = float((gr - grr).abs().max() / grr.abs().max())
        print("%-28s %s backward %.4f ms (min %.4f)  frac %.3f  dev vs natural-order %.2e  status %d" % (tag, str(dt)[6:], b, bmin, by / b / 1e6 / 8000, err, int(st.abs().sum())), flush=True)
        if dt == torch.float64:
            # float64 in, float32 gradient out: what the reference's mlpg_grad returns (_mlpg.py:248)
            b32, b32min = timeit(lambda: _hip.backward(v, g, W3, 3 * sd, out_dtype=torch.float32, algo=3, want_status=False))
            gr, _ = _hip.backward(v[:8], g[:8], W3, 3 * sd, out_dtype=torch.float32, algo=3)
            err = float((gr.double() - grr).abs().max() / grr.abs().max())
            by32 = (8 * 4 + 4 * 3) * sd * B * T
            print("%-28s float64 -> float32 backward %.4f ms (min %.4f)  frac %.3f  dev vs natural-order (float64) %.2e" % (tag, b32, b32min, by32 / b32 / 1e6 / 8000, err), flush=True)
