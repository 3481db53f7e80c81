"""Host-memory calls on arrays that are NOT ordinary anonymous memory: read-only np.memmap inputs (np.load(..., mmap_mode="r")), a
writable memmap as the result array of the C entry point, arrays in POSIX shared memory -- sizes that take the runtime's direct copy
(>= 4 MB per array).  Every result against the same call on ordinary copies."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nnmnkwii_amd import _hip  # noqa: E402
from nnmnkwii_amd import paramgen as G  # noqa: E402

W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
B, T, sd = 6, 1000, 60
M_ = rng.randn(B, T, 3 * sd)
V_ = rng.rand(B, T, 3 * sd) + 0.1
want = G.mlpg_batch(M_.copy(), V_.copy(), W)
d = tempfile.mkdtemp(dir="/tmp")
np.save(os.path.join(d, "m.npy"), M_)
np.save(os.path.join(d, "v.npy"), V_)
Mr = np.load(os.path.join(d, "m.npy"), mmap_mode="r")
Vr = np.load(os.path.join(d, "v.npy"), mmap_mode="r")
print("read-only memmap inputs (%s, writeable=%s): " % (type(Mr).__name__, Mr.flags.writeable), end="")
for k in range(2):
    y = G.mlpg_batch(Mr, Vr, W)
    assert np.array_equal(y, want)
print("ok (twice)")
go = rng.randn(B, T, sd)
gw, _ = _hip.backward_host(V_.copy(), go.copy(), W, 3 * sd, out_dtype=np.float64)
g, _ = _hip.backward_host(np.ascontiguousarray(Vr), go, W, 3 * sd, out_dtype=np.float64)
assert np.array_equal(g, gw)
print("backward on the memmapped variances: ok")
# a writable, file-backed result array handed to the C entry point
L = _hip.lib()
out = np.lib.format.open_memmap(os.path.join(d, "y.npy"), mode="w+", dtype=np.float64, shape=(B, T, sd))
st = np.zeros((B, sd), dtype=np.int32)
pw = _hip.cached_windows(W)
pl, pu, pc = pw.ptrs()
for k in range(2):
    out[...] = 0
    rc = L.mlpg_hip_forward_host(0, _hip.F64, 0, Mr.ctypes.data, Vr.ctypes.data, _hip.VAR_FRAME, None, B, T, 3 * sd, 3, pl, pu, pc, out.ctypes.data, st.ctypes.data)
    assert rc == 0, _hip.lib().mlpg_hip_last_error()
    assert np.array_equal(np.asarray(out), want)
print("file-backed result array: ok (twice)")
from multiprocessing import shared_memory  # noqa: E402
shm = shared_memory.SharedMemory(create=True, size=M_.nbytes)
Ms = np.ndarray(M_.shape, dtype=M_.dtype, buffer=shm.buf)
Ms[...] = M_
for k in range(2):
    assert np.array_equal(G.mlpg_batch(Ms, V_, W), want)
print("POSIX shared memory input: ok (twice)")
del Ms
shm.close()
shm.unlink()
