"""In-kernel timeline of the strip kernel (library built with -DMLPG_STRIP_TRACE: every item's start, loads-landed, arrived,
level-3-done and end times on the 100 MHz constant clock, plus the XCD it ran on, dumped through the status array) at the
config-2 shape: phase lengths, how many workgroups are loading at any instant -- over the whole device and per XCD -- and
whether the workgroups of an XCD load in lockstep (oscillation of the per-XCD loading count).
usage: NNMNKWII_AMD_SO=.../libmlpg_hip_trace.so python tools/dbg/strip_trace.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402

W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
B, T, sd = 256, 1000, 60
gen = torch.Generator(device="cuda").manual_seed(1234)
m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen)
v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda", generator=gen) + 0.1
for _ in range(60):
    y, st = _hip.forward(m, v, W3, None, algo=3, want_status=True)
torch.cuda.synchronize()
R = (T + 63) // 64
tr = st.cpu().numpy()[:B * R * 4].reshape(-1, 4).astype(np.int64)
t0 = tr[:, 0]
t0 = (t0 - t0.min()) & 0x3FFFFFFF
loads, arrived = tr[:, 1] & 0xFFFF, (tr[:, 1] >> 16) & 0xFFFF
l3, end, xcd = tr[:, 2], tr[:, 3] & 0xFFFFFF, (tr[:, 3] >> 24) & 7
print("items %d; ticks of 10 ns" % len(t0))
for nm, a in (("start -> loads landed", loads), ("loads landed -> neighbours arrived", arrived - loads), ("arrived -> level 3 done", l3 - arrived),
              ("level 3 done -> end", end - l3), ("whole item", end)):
    print("  %-36s median %5d  p10 %5d  p90 %5d  mean %6.1f" % (nm, np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.mean()))
span = int((t0 + end).max()) + 1
print("kernel span %d ticks = %.1f us" % (span, span / 100.0))


def series(mask):
    ld = np.zeros(span + 1)
    ac = np.zeros(span + 1)
    np.add.at(ld, t0[mask], 1)
    np.add.at(ld, (t0 + loads)[mask], -1)
    np.add.at(ac, t0[mask], 1)
    np.add.at(ac, (t0 + end)[mask], -1)
    return np.cumsum(ld)[:span], np.cumsum(ac)[:span]


ld, ac = series(np.ones(len(t0), bool))
body = slice(int(0.1 * span), int(0.85 * span))
print("device: workgroups loading mean %.0f (std %.0f), active mean %.0f, in the middle 75 %% of the launch" % (ld[body].mean(), ld[body].std(), ac[body].mean()))
nb = span // 100
print("  loading per 1 us:", [int(x) for x in ld[:nb * 100].reshape(nb, 100).mean(1)])
for x in range(8):
    lx, ax = series(xcd == x)
    s = lx[body]
    # dominant oscillation period of the loading count (autocorrelation peak between 5 and 40 us)
    z = s - s.mean()
    ac_ = np.correlate(z[::10], z[::10], "full")[len(z[::10]) - 1:]
    ac_ = ac_ / max(ac_[0], 1e-9)
    lag = 50 + int(np.argmax(ac_[50:400])) if len(ac_) > 400 else -1
    print("  XCD %d: items %4d  loading mean %5.1f std %5.1f (of %d resident)  active mean %5.1f  autocorrelation peak %.2f at %.1f us"
          % (x, int((xcd == x).sum()), s.mean(), s.std(), 64, ax[body].mean(), ac_[lag] if lag > 0 else 0.0, lag / 10.0))
print("  XCD 0 loading per 1 us:", [int(x) for x in series(xcd == 0)[0][:nb * 100].reshape(nb, 100).mean(1)])
