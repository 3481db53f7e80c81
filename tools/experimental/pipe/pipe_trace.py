"""Per-item realtime stamps of the pipelined kernel (build with -DMLPG_PIPE_TRACE): who waits for whom."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
from tools.bench_paths import WINDOWS
B, T, sd = 256, 1000, 60
m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device="cuda")
v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device="cuda") + 0.1
for _ in range(2):
    out, st = _hip.forward(m, v, WINDOWS, algo=_hip.ALGO_PIPE)
torch.cuda.synchronize()
s = st.cpu().numpy()[:32 * 24 * 10].reshape(32, 24, 10).astype(np.int64)
valid = s > 0
t0 = s[valid].min()
names = ["body0", "body1", "tkt", "u", "fin", "recseen", "pub", "arr", "l3", "post"]
np.set_printoptions(linewidth=250)
print("span of all stamps: %.1f us" % ((s[valid].max() - t0) / 100.0))
for wg in (3, 17):
    print("workgroup", wg, "(10 ns ticks since the first stamp; columns:", " ".join(names), ")")
    for it in range(0, 23):
        row = s[wg, it]
        if not row.any():
            continue
        print("  item %2d: " % it + " ".join("%6d" % ((x - t0) if x else -1) for x in row))
def d(a, b):
    x = s[:, 2:19, b] - s[:, 2:19, a]
    ok = (s[:, 2:19, a] > 0) & (s[:, 2:19, b] > 0)
    return x[ok].mean()
print("means over items 2..18, 32 workgroups (ticks of 10 ns):")
print("  body %.0f | body end -> next ticket %.0f | ticket -> u seen (prologue issue + wait) %.0f | backsub+stores %.0f | records -> seen by chain %.0f | level2+publish %.0f | published -> arrived %.0f | level 3 %.0f | post %.0f"
      % (d(0, 1), d(1, 2), 0, d(3, 4), d(1, 5), d(5, 6), d(6, 7), d(7, 8), d(8, 9)))
x = s[:, 3:19, 0] - s[:, 2:18, 0]
ok = (s[:, 3:19, 0] > 0) & (s[:, 2:18, 0] > 0)
print("  records written -> separators posted %.0f ; period (body start to next body start) %.0f" % (d(1, 9), x[ok].mean()))
# when do workgroups start their first and end their last item
first = np.where(s[:, 0, 0] > 0, s[:, 0, 0], s[:, 1, 0]) - t0
last = s[:, :, 4].max(axis=1) - t0
print("  first body start per workgroup: min %d max %d ; last finish: min %d max %d" % (first.min(), first.max(), last.min(), last.max()))
