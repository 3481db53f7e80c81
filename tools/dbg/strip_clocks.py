"""What shader clock does the device run at while the strip kernel saturates it?  A loop of config-2 launches for a few seconds, rocm-smi
sampled meanwhile (sclk / mclk / power), against the same with a launch of 8 utterances only (8 workgroups: the device almost idle)."""
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
gen = torch.Generator(device="cuda").manual_seed(1)
m = torch.randn(256, 1000, 180, dtype=torch.float64, device="cuda", generator=gen)
v = torch.rand(256, 1000, 180, dtype=torch.float64, device="cuda", generator=gen) + 0.1


def smi():
    out = []
    for cmd in (["rocm-smi", "--showclocks", "--showpower"], ["amd-smi", "metric", "-c", "-p"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if r.returncode == 0 and r.stdout.strip():
                out.append(r.stdout)
                break
        except Exception as e:  # noqa: BLE001
            out.append("%s: %s" % (cmd[0], e))
    return "\n".join(out)


def load(B, seconds, samples):
    stop = [False]

    def sampler():
        time.sleep(seconds * 0.4)
        samples.append(smi())
        time.sleep(seconds * 0.3)
        samples.append(smi())

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(200):
            _hip.forward(m[:B], v[:B], W3, None, algo=3, want_status=False)
        torch.cuda.synchronize()
        n += 200
    th.join()
    return n / (time.time() - t0)


print("== idle"); print(smi()[:1500])
for B in (256, 8):
    s = []
    rate = load(B, 6.0, s)
    print("== %d utterances per launch, %.0f launches/s" % (B, rate))
    for x in s:
        lines = [l for l in x.splitlines() if any(k in l.lower() for k in ("sclk", "mclk", "fclk", "socclk", "power", "gfx", "clk"))]
        print("\n".join(lines[:14]))
        print("--")
