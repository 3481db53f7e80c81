"""The transposed strip form against the wave-per-system kernel on narrow streams WITH a lengths vector: full lengths, mildly ragged
(uniform in [0.8 T, T], as length-bucketed batches are) and uniform in [1, T] (the worst case: a lane group runs to its longest utterance).
usage: python tools/dbg/narrow_ragged_time.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import gpu_time

W3 = _hip.prepack_windows([(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))])
gen = torch.Generator(device="cuda").manual_seed(1)
rng = np.random.RandomState(0)
for B, T, sd in [(512, 2000, 1), (512, 2000, 5), (256, 1000, 5), (256, 1000, 25)]:
    m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen)
    v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen) + 0.1
    for name, lens in (("full", np.full(B, T)), ("bucketed [0.8 T, T]", rng.randint(int(0.8 * T), T + 1, size=B)), ("uniform [1, T]", rng.randint(1, T + 1, size=B))):
        L = torch.from_numpy(lens.astype(np.int32)).cuda()
        tw = gpu_time(lambda: _hip.forward(m, v, W3, L, algo=_hip.ALGO_WAVE, want_status=False), steps=20, warmup=3)
        tt = gpu_time(lambda: _hip.forward(m, v, W3, L, algo=_hip.ALGO_STRIP, want_status=False), steps=20, warmup=3)
        n0 = int(_hip.lib().mlpg_hip_launch_count(9))
        _hip.forward(m, v, W3, L, want_status=False)
        auto_tr = int(_hip.lib().mlpg_hip_launch_count(9)) > n0
        print("B=%d T=%d sd=%d  lengths %-20s mean %.0f:  wave %.4f  transposed %.4f  (AUTO takes the transposed form: %s)" % (B, T, sd, name, lens.mean(), tw, tt, auto_tr), flush=True)
