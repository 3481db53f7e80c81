"""Time the constant-coefficient kernel against the wave / strip kernels on the global- and unit-variance shapes
(config 2 with (D,) variances, config 3 forward / backward).  HIP events on the launch stream, median of reps."""
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
    for _ in range(reps):      # queued ahead: the host's launch latency must not sit between the events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = [e0.elapsed_time(e1) for e0, e1 in evs]
    return float(np.median(ts)), float(np.min(ts))


def main():
    torch.manual_seed(0)
    names = {2: "wave", 3: "strip", 5: "const"}
    for (B, T, sd, dt, tag) in [(256, 1000, 60, torch.float64, "c2g f64"), (256, 1000, 60, torch.float32, "c2g f32"),
                                (64, 500, 60, torch.float32, "c3 f32"), (64, 4000, 60, torch.float64, "T4000 f64"),
                                (512, 2000, 60, torch.float64, "c5 mgc f64")]:
        m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
        vg = torch.rand(3 * sd, dtype=dt, device="cuda") + 0.1
        go = torch.randn(B, T, sd, dtype=dt, device="cuda")
        esz = m.element_size()
        for mode, var in (("global", vg), ("unit", None)):
            for algo in (2, 3, 5):
                if algo == 2 and T > 2048:
                    continue
                try:
                    med, mn = timeit(lambda: _hip.forward(m, var, W3, None, algo=algo, want_status=False))
                    byts = B * T * sd * 4 * esz
                    print("%-10s fwd %-6s %-5s  %.4f ms (min %.4f)  %.0f GB/s  frac %.3f" % (tag, mode, names[algo], med, mn, byts / med / 1e6, byts / med / 1e6 / 8000), flush=True)
                except Exception as ex:
                    print(tag, mode, names[algo], "ERR", ex)
            for algo in (2, 3, 5):
                if algo == 2 and T > 2048:
                    continue
                try:
                    med, mn = timeit(lambda: _hip.backward(var, go, W3, 3 * sd, None, out_dtype=dt, algo=algo, want_status=False))
                    byts = B * T * sd * 4 * esz
                    print("%-10s bwd %-6s %-5s  %.4f ms (min %.4f)  %.0f GB/s  frac %.3f" % (tag, mode, names[algo], med, mn, byts / med / 1e6, byts / med / 1e6 / 8000), flush=True)
                except Exception as ex:
                    print(tag, mode, names[algo], "ERR", ex)


if __name__ == "__main__":
    main()
