"""Narrow streams (1 .. 32 static dims, per-frame variances, no lengths): the strip kernel's transposed form (MLPG_HIP_ALGO_STRIP takes it
where it applies) against the wave-per-system kernel, and what AUTO picks.  HIP events around each call, median of 20.
usage: python tools/dbg/narrow_time.py [frame|global|unit]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import gpu_time

W3 = _hip.prepack_windows([(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))])
SHAPES = [(512, 2000, 1), (512, 2000, 2), (512, 2000, 5), (256, 1000, 1), (256, 1000, 5), (256, 1000, 16), (256, 1000, 25), (256, 1000, 32),
          (64, 500, 1), (64, 500, 5), (512, 200, 1), (512, 200, 5), (128, 300, 8), (32, 2000, 3), (1024, 256, 1)]
gen = torch.Generator(device="cuda").manual_seed(1)
VM = sys.argv[1] if len(sys.argv) > 1 else "frame"      # frame | global | unit
for dt in (torch.float64, torch.float32):
    for B, T, sd in SHAPES:
        m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda", generator=gen)
        v = (torch.rand(B, T, 3 * sd, dtype=dt, device="cuda", generator=gen) + 0.1 if VM == "frame"
             else torch.rand(3 * sd, dtype=dt, device="cuda", generator=gen) + 0.5 if VM == "global" else None)
        res = {}
        for name, algo in (("auto", _hip.ALGO_AUTO), ("wave", _hip.ALGO_WAVE), ("strip(tr)", _hip.ALGO_STRIP)) + ((("const", _hip.ALGO_CONST),) if VM != "frame" else ()):
            try:
                _hip.forward(m, v, W3, algo=algo, want_status=False)
                torch.cuda.synchronize()
                n0 = int(_hip.lib().mlpg_hip_launch_count(9))
                res[name] = gpu_time(lambda: _hip.forward(m, v, W3, algo=algo, want_status=False), steps=20, warmup=3)
                if name == "auto":
                    res["auto_is_tr"] = int(_hip.lib().mlpg_hip_launch_count(9)) > n0
            except Exception as e:   # noqa: BLE001
                res[name] = None
        u = 64 // sd
        items = -(-B // u) * -(-T // 64)
        by = (8 if dt == torch.float64 else 4) * (7 if VM == "frame" else 4) * sd * B * T
        print("%s %s B=%-5d T=%-5d sd=%-3d items=%-6d  auto %s (tr=%s)  wave %s  strip(tr) %s   tr frac %.3f%s" % (
            VM, str(dt)[6:], B, T, sd, items, "%.4f" % res["auto"], res.get("auto_is_tr"), "%.4f" % res["wave"] if res["wave"] else None,
            "%.4f" % res["strip(tr)"] if res["strip(tr)"] else None, by / res["strip(tr)"] / 1e6 / 8000 if res["strip(tr)"] else 0,
            ("  const %.4f" % res["const"]) if res.get("const") else ""), flush=True)
        del m, v
