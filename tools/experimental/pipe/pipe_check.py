"""Quick GPU check of the pipelined kernel against the generic one over a few shapes (run under `timeout`)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS, gpu_time

torch.manual_seed(0)
ok = True
shapes = [(1, 40, 3), (2, 48, 60), (2, 49, 60), (3, 100, 60), (4, 200, 70), (3, 1000, 60), (8, 1000, 130), (2, 4100, 20), (256, 1000, 60)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for B, T, sd in shapes:
    for dt in (torch.float64, torch.float32):
        m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
        v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.1
        L = torch.randint(1, T + 1, (B,), dtype=torch.int32, device="cuda")
        L[0] = T
        for name, var, lens in (("frame", v, L), ("frame-full", v, None), ("global", v[0, 0].contiguous(), L), ("unit", None, L)):
            ref, _ = _hip.forward(m, var, WINDOWS, lens, algo=_hip.ALGO_GENERIC)
            out, st = _hip.forward(m, var, WINDOWS, lens, algo=_hip.ALGO_PIPE)
            torch.cuda.synchronize()
            scale = float(ref.abs().max()) + 1e-300
            err = float((out - ref).abs().max()) / scale
            bad = int(st.abs().max())
            tol = 1e-9 if dt == torch.float64 else 5e-6
            flag = "ok" if (err <= tol and bad == 0) else "FAIL"
            ok &= flag == "ok"
            print("fwd B=%d T=%d sd=%d %s %s: err %.2e status %d %s" % (B, T, sd, str(dt)[6:], name, err, bad, flag), flush=True)
        go = torch.randn(B, T, sd, dtype=dt, device="cuda")
        ref, _ = _hip.backward(v, go, WINDOWS, 3 * sd, L, out_dtype=dt, algo=_hip.ALGO_GENERIC)
        out, st = _hip.backward(v, go, WINDOWS, 3 * sd, L, out_dtype=dt, algo=_hip.ALGO_PIPE)
        torch.cuda.synchronize()
        err = float((out - ref).abs().max()) / (float(ref.abs().max()) + 1e-300)
        flag = "ok" if (err <= (1e-9 if dt == torch.float64 else 5e-6) and int(st.abs().max()) == 0) else "FAIL"
        ok &= flag == "ok"
        print("bwd B=%d T=%d sd=%d %s: err %.2e %s" % (B, T, sd, str(dt)[6:], err, flag), flush=True)
print("ALL OK" if ok else "FAILURES")
B, T, sd = 256, 1000, 60
m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda")
v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda") + 0.1
for name, algo in (("strip", _hip.ALGO_STRIP), ("pipe", _hip.ALGO_PIPE)):
    ms = gpu_time(lambda: _hip.forward(m, v, WINDOWS, algo=algo, want_status=False), steps=30, warmup=5)
    print("config 2 forward f64 %s: %.4f ms  frac %.3f" % (name, ms, 56.0 * sd * B * T / ms / 1e6 / 8000), flush=True)
