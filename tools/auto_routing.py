#!/usr/bin/env python
"""What MLPG_HIP_ALGO_AUTO picks, and what that costs: for a list of representative launches (shape, dtype, variance mode,
window set, direction) the kernel family AUTO routes to -- read off the library's launch counters, not inferred -- and the
time of AUTO and of every explicit algorithm that accepts the launch.

    python tools/auto_routing.py                 # on the GPU box: prints the markdown table, writes gpurun_out/auto_routing.json
    python tools/auto_routing.py --routes-only   # routes without timings (what tests/test_auto_routing_gpu.py re-derives)

The committed copy (profiles/r05_auto_routing.json) is the table of DESIGN.md section "AUTO routing" and the expectation of
tests/test_auto_routing_gpu.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

KINDS = ["generic", "wave", "strip", "strip_multi", "const", "fused", "chunk", "fir", "const_multi", "strip_tr"]   # csrc/common.h kCount*
ALGOS = {"generic": 1, "wave": 2, "strip": 3, "const": 5, "chunk": 6, "fir": 7}          # include/mlpg_hip.h MLPG_HIP_ALGO_*

# (name, B, T, sd, dtype, variance mode, window set, direction)
CASES = [
    ("c2 per-frame variances f64", 256, 1000, 60, "f64", "frame", "std3", "fwd"),
    ("c2 per-frame variances f64", 256, 1000, 60, "f64", "frame", "std3", "bwd"),
    ("c2 per-frame variances f32", 256, 1000, 60, "f32", "frame", "std3", "fwd"),
    ("c2 per-frame variances f32", 256, 1000, 60, "f32", "frame", "std3", "bwd"),
    ("c2 global (D,) variances f64", 256, 1000, 60, "f64", "global", "std3", "fwd"),
    ("c2 global (D,) variances f64", 256, 1000, 60, "f64", "global", "std3", "bwd"),
    ("c2 unit variances f64", 256, 1000, 60, "f64", "unit", "std3", "fwd"),
    ("c2 unit variances f32", 256, 1000, 60, "f32", "unit", "std3", "fwd"),
    ("c2 unit variances f32", 256, 1000, 60, "f32", "unit", "std3", "bwd"),
    ("c3 unit variances f32", 64, 500, 60, "f32", "unit", "std3", "fwd"),
    ("c3 unit variances f32", 64, 500, 60, "f32", "unit", "std3", "bwd"),
    ("c3 per-frame variances f32", 64, 500, 60, "f32", "frame", "std3", "fwd"),
    ("c5 mgc per-frame f64", 512, 2000, 60, "f64", "frame", "std3", "fwd"),
    ("c5 mgc global f64", 512, 2000, 60, "f64", "global", "std3", "fwd"),
    ("c5 lf0 (1 dim) f64", 512, 2000, 1, "f64", "frame", "std3", "fwd"),
    ("c5 bap (5 dims) f64", 512, 2000, 5, "f64", "frame", "std3", "fwd"),
    ("25 dims f64", 256, 1000, 25, "f64", "frame", "std3", "fwd"),
    ("128 dims f64", 256, 1000, 128, "f64", "frame", "std3", "fwd"),
    ("long utterances f64", 64, 4000, 60, "f64", "frame", "std3", "fwd"),
    ("one utterance f64", 1, 1000, 60, "f64", "frame", "std3", "fwd"),
    ("short utterances f64", 256, 100, 60, "f64", "frame", "std3", "fwd"),
    ("two windows f64", 256, 1000, 60, "f64", "frame", "std2", "fwd"),
    ("5-tap windows f64", 256, 1000, 60, "f64", "frame", "wide3", "fwd"),
    ("5-tap windows f64", 256, 1000, 60, "f64", "frame", "wide3", "bwd"),
    ("config 1 (T=100, 2 dims) f64", 1, 100, 2, "f64", "frame", "std3", "fwd"),
    ("lf0 of a config-2 batch f64", 256, 1000, 1, "f64", "frame", "std3", "fwd"),
    ("bap (5 dims) of a config-2 batch f32", 256, 1000, 5, "f32", "frame", "std3", "fwd"),
    ("16 dims f64", 256, 1000, 16, "f64", "frame", "std3", "fwd"),
    ("lf0, short utterances f64", 512, 200, 1, "f64", "frame", "std3", "fwd"),
]


def make(case, torch):
    from cases import WINDOW_SETS
    name, B, T, sd, dt, vm, wname, direction = case
    win = WINDOW_SETS[wname]
    nw = len(win)
    tdt = torch.float64 if dt == "f64" else torch.float32
    gen = torch.Generator(device="cuda").manual_seed(7)
    m = torch.randn(B, T, nw * sd, dtype=tdt, device="cuda", generator=gen)
    if vm == "frame":
        v = torch.rand(B, T, nw * sd, dtype=tdt, device="cuda", generator=gen) + 0.5
    elif vm == "global":
        v = torch.rand(nw * sd, dtype=tdt, device="cuda", generator=gen) + 0.5
    else:
        v = None
    g = torch.randn(B, T, sd, dtype=tdt, device="cuda", generator=gen)
    return win, nw, m, v, g, tdt


def runner(case, torch, _hip):
    win, nw, m, v, g, tdt = make(case, torch)
    pw = _hip.prepack_windows(win)
    sd = case[3]
    if case[7] == "fwd":
        return lambda algo: _hip.forward(m, v, pw, algo=algo, want_status=False)
    return lambda algo: _hip.backward(v, g, pw, nw * sd, out_dtype=tdt, algo=algo, want_status=False)


def route_of(fn, _hip):
    L = _hip.lib()
    before = [int(L.mlpg_hip_launch_count(k)) for k in range(len(KINDS))]
    fn(_hip.ALGO_AUTO)
    after = [int(L.mlpg_hip_launch_count(k)) for k in range(len(KINDS))]
    hit = [KINDS[k] for k in range(len(KINDS)) if after[k] != before[k]]
    return "+".join(hit) if hit else "none"


def main():
    import torch
    from nnmnkwii_amd import _hip
    from tools.bench_paths import gpu_time
    routes_only = "--routes-only" in sys.argv
    rows = []
    for case in CASES:
        fn = runner(case, torch, _hip)
        fn(_hip.ALGO_AUTO)
        torch.cuda.synchronize()
        row = {"case": case[0], "B": case[1], "T": case[2], "sd": case[3], "dtype": case[4], "variances": case[5],
               "windows": case[6], "direction": case[7], "auto": route_of(fn, _hip)}
        if not routes_only:
            row["ms_auto"] = gpu_time(lambda: fn(_hip.ALGO_AUTO), steps=20, warmup=3)
            row["ms"] = {}
            for aname, algo in ALGOS.items():
                if aname == "generic" and case[1] * case[2] * case[3] > 4e6:
                    continue      # (the natural-order kernel takes milliseconds at the big shapes: not a candidate there)
                try:
                    fn(algo)
                    torch.cuda.synchronize()
                    row["ms"][aname] = gpu_time(lambda: fn(algo), steps=12, warmup=2)
                except Exception:   # noqa: BLE001 -- the kernel does not accept this launch
                    pass
        rows.append(row)
        print(json.dumps(row), flush=True)
        del fn
        torch.cuda.empty_cache()
    if not routes_only:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "auto_routing.json"), "w") as f:
            json.dump(rows, f, indent=1)
        print(markdown(rows))


def markdown(rows):
    out = ["| launch | B × T × dims | dtype | variances | windows | dir | AUTO → | AUTO ms | every accepting kernel, ms |",
           "|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        alts = ", ".join("%s %.4f" % (k, v) for k, v in sorted(r.get("ms", {}).items(), key=lambda kv: kv[1]))
        out.append("| %s | %d × %d × %d | %s | %s | %s | %s | **%s** | %.4f | %s |" % (
            r["case"], r["B"], r["T"], r["sd"], r["dtype"], r["variances"], r["windows"], r["direction"], r["auto"],
            r.get("ms_auto", float("nan")), alts))
    return "\n".join(out)


if __name__ == "__main__":
    if "--markdown" in sys.argv:
        print(markdown(json.load(open(sys.argv[sys.argv.index("--markdown") + 1]))))
    else:
        main()
