#!/usr/bin/env python
"""Forward MLPG kernel time per algorithm over a grid of shapes (1 GPU): the data behind the AUTO policy of
mlpg_hip_forward (capi.hip: dispatch_solve).  One line per shape: ms for wave / strip / generic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_paths import WINDOWS, gpu_time  # noqa: E402


def main():
    import torch
    from nnmnkwii_amd import _hip
    backward = "--backward" in sys.argv
    shapes = [(256, 1000, 60), (64, 1000, 60), (16, 1000, 60), (4, 1000, 60), (1, 1000, 60), (256, 300, 60), (64, 300, 60),
              (256, 1000, 25), (256, 1000, 16), (256, 1000, 8), (256, 1000, 5), (512, 2000, 5), (512, 2000, 1),
              (64, 500, 60), (32, 2000, 60), (8, 4000, 60), (256, 100, 60), (1024, 100, 60), (256, 1000, 64), (256, 1000, 80), (256, 1000, 128)]
    if backward:
        shapes = [(256, 1000, 60), (64, 1000, 60), (256, 300, 60), (64, 500, 60), (32, 2000, 60), (256, 1000, 25), (256, 1000, 128)]
    for dt in (torch.float64, torch.float32):
        for B, T, sd in shapes:
            m = torch.randn(B, T, 3 * sd, dtype=dt, device="cuda")
            v = torch.rand(B, T, 3 * sd, dtype=dt, device="cuda") + 0.5
            pw = _hip.prepack_windows(WINDOWS)
            row = []
            for algo in (_hip.ALGO_WAVE, _hip.ALGO_STRIP, _hip.ALGO_GENERIC, _hip.ALGO_CONST):
                try:
                    if backward:
                        go = m[:, :, :sd].contiguous()
                        ms = gpu_time(lambda: _hip.backward(v, go, pw, 3 * sd, out_dtype=dt, algo=algo, want_status=False), steps=20, warmup=3)
                    else:
                        ms = gpu_time(lambda: _hip.forward(m, v, pw, algo=algo), steps=20, warmup=3)
                except Exception as e:  # unsupported shape for that kernel
                    ms = float("nan")
                row.append(ms)
            best = ["wave", "strip", "generic", "const"][int(np.nanargmin(row))]
            print(("bwd " if backward else "fwd ") + "%s B=%4d T=%4d sd=%3d  wave %.4f  strip %.4f  generic %.4f  const %.4f  -> %s" % (str(dt)[6:], B, T, sd, row[0], row[1], row[2], row[3], best), flush=True)
            del m, v


if __name__ == "__main__":
    main()
