"""Does the NUMA placement of the calling process matter for the host-memory calls?  Finds the GPU's NUMA node
(/sys/bus/pci/devices/<bdf>/numa_node), then times the literal calls with the process pinned (os.sched_setaffinity, memory
allocated after pinning: first touch) to the GPU's node, to every other node in turn, and unpinned.
usage: python tools/dbg/numa_affinity.py"""
import ctypes
import glob
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


L = _hip.lib()
buf = ctypes.create_string_buffer(64)
rc = L.hipDeviceGetPCIBusId(buf, 64, 0)       # (dlsym on the library's handle also searches the HIP runtime it was linked against)
bdf = buf.value.decode().lower()
node_path = "/sys/bus/pci/devices/%s/numa_node" % bdf
gpu_node = int(open(node_path).read()) if os.path.exists(node_path) else -1
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(os.path.basename(d)[4:])] = cpulist(open(d + "/cpulist").read())
print("GPU 0 is %s (hipDeviceGetPCIBusId rc %d), NUMA node %d; nodes: %s; this process may run on %d cpus"
      % (bdf, rc, gpu_node, {k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items()}, len(os.sched_getaffinity(0))))
allowed = os.sched_getaffinity(0)


def measure(tag):
    rng = np.random.RandomState(0)
    res = []
    for T, sd, n in ((100, 2, 400), (1000, 60, 200)):
        m = rng.randn(T, 3 * sd)
        v = rng.rand(T, 3 * sd) + 0.1
        for _ in range(10):
            G.mlpg(m, v, W)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            G.mlpg(m, v, W)
            ts.append(time.perf_counter() - t0)
        res.append("T=%d sd=%d: %.1f us (min %.1f)" % (T, sd, np.median(ts) * 1e6, min(ts) * 1e6))
    B = 32
    M_ = rng.randn(B, 1000, 180)
    V_ = rng.rand(B, 1000, 180) + 0.1
    for _ in range(2):
        G.mlpg_batch(M_, V_, W)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        G.mlpg_batch(M_, V_, W)
        ts.append(time.perf_counter() - t0)
    res.append("batch of 32 (92 MB): %.2f ms (min %.2f)" % (np.median(ts) * 1e3, min(ts) * 1e3))
    # config 3's eager training loop (host-bound: a dozen launches per step), wall clock per step
    import torch
    from nnmnkwii_amd import autograd as AF
    dev = torch.device("cuda:0")
    R = torch.from_numpy(G.unit_variance_mlpg_matrix(W, 500)).to(dev)
    means = torch.rand(64, 500, 180, device=dev, requires_grad=True)
    y = torch.rand(64, 500, 60, device=dev)
    crit = torch.nn.MSELoss()

    def step():
        means.grad = None
        crit(AF.unit_variance_mlpg(R, means), y).backward()

    for _ in range(100):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(1000):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1000 * 1e6)
    res.append("config-3 eager loop: %.1f us per step" % best)
    print("%-34s %s" % (tag, "; ".join(res)), flush=True)


measure("unpinned (as launched)")
for k, cpus in nodes.items():
    use = set(cpus) & allowed
    if not use:
        continue
    os.sched_setaffinity(0, use)
    _hip.lib().mlpg_hip_shutdown()       # (the helper threads are recreated under the new mask, the pinned buffers reallocated: first touch)
    measure("pinned to node %d%s" % (k, " (the GPU's)" if k == gpu_node else ""))
os.sched_setaffinity(0, allowed)
_hip.lib().mlpg_hip_shutdown()
measure("unpinned again")
