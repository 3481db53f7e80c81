"""GPU tests (-m gpu) of the host-memory entry points over a device LIST (mlpg_hip_forward_host_multi,
mlpg_hip_fastdtw_host_multi): one process deals the utterance / pair chunks round-robin over the listed devices.  A
one-GPU box lists its device several times -- each occurrence has its own streams and staging buffers, so the dealing,
the slot reuse and the positional merge of outputs and verdicts run exactly as they do over distinct GPUs."""
import numpy as np
import pytest

from cases import WINDOW_SETS
from oracle import dtw as OD
from oracle import mlpg as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], "all"])
def test_forward_host_over_a_device_list_against_the_oracle(devices):
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import paramgen as G
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(31)
    B, T, sd = 53, 200, 40           # 53 utterances: 4 n chunks, the last one short
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
    y1, st1 = _hip.forward_host(M_, V_, windows, lengths)                 # the single-device call
    y, st = _hip.forward_host(M_, V_, windows, lengths, device=devices)
    assert np.array_equal(y, y1) and np.array_equal(st, st1) and not st.any()
    yo, _, rc = O.mlpg_batch(M_, V_, windows, lengths)
    assert rc == 0
    sc = np.abs(yo).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(y - yo) / sc).max() <= 1e-9
    # global variances (sent once per stream slot), unit variances, float32, through the drop-in batch call
    vg = V_[0, 0].copy()
    yg = G.mlpg_batch(M_, vg, windows, lengths, device=devices)
    ygo, _, _ = O.mlpg_batch(M_, vg, windows, lengths)
    assert (np.abs(yg - ygo) / (np.abs(ygo).max(axis=1, keepdims=True) + 1e-300)).max() <= 1e-9
    yu = G.mlpg_batch(M_.astype(np.float32), None, windows, lengths, device=devices)
    yuo, _, _ = O.mlpg_batch(M_.astype(np.float32), np.ones(3 * sd, dtype=np.float32), windows, lengths)
    assert (np.abs(yu - yuo) / (np.abs(yuo).max(axis=1, keepdims=True) + 1e-300)).max() <= 5e-6
    # verdicts merge by position: a bad system in a chunk that another list entry ran
    Vb = V_.copy()
    Vb[41, 0, 7] = -1e-3
    _, stb = _hip.forward_host(M_, Vb, windows, None, device=devices)
    _, sto, _ = O.mlpg_batch(M_[41:42], Vb[41:42], windows)
    bad = np.argwhere(stb != 0)
    assert bad.tolist() == [[41, 7]] and stb[41, 7] == sto[0, 7]


def test_fastdtw_host_over_a_device_list_against_the_oracle():
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    rng = np.random.RandomState(32)
    N, Tx, Ty, D = 21, 90, 110, 5
    X = np.zeros((N, Tx, D))
    Y = np.zeros((N, Ty, D))
    lx = rng.randint(20, Tx + 1, N)
    ly = rng.randint(20, Ty + 1, N)
    for n in range(N):
        X[n, :lx[n]] = np.cumsum(rng.randn(lx[n], D), 0) * 0.1 + 2.0
        Y[n, :ly[n]] = np.cumsum(rng.randn(ly[n], D), 0) * 0.1 + 2.0
    one = _hip.fastdtw_host(X, Y, 1)
    for devices in ([0, 0], [0, 0, 0, 0], "all"):
        many = _hip.fastdtw_host(X, Y, 1, device=devices)
        for a, b in zip(one[2:], many[2:]):                     # path_len, cost, lenx, leny
            assert np.array_equal(a, b)
        for n in range(N):                                      # (path slots past path_len are unspecified)
            k = one[2][n]
            assert np.array_equal(one[0][n, :k], many[0][n, :k]) and np.array_equal(one[1][n, :k], many[1][n, :k])
    pi, pj, pl, cost, hx, hy = one
    assert hx.tolist() == lx.tolist() and hy.tolist() == ly.tolist()
    for n in range(N):
        d, path = OD.fastdtw(X[n, :lx[n]], Y[n, :ly[n]], 1)
        assert pl[n] == len(path) and np.array_equal(pi[n, :pl[n]], path[:, 0]) and np.array_equal(pj[n, :pl[n]], path[:, 1])
    Xa, Ya = DTWAligner().transform((X, Y))
    Xb, Yb = DTWAligner(devices=[0, 0]).transform((X, Y))
    assert np.array_equal(Xa, Xb) and np.array_equal(Ya, Yb)


def test_device_list_errors():
    from nnmnkwii_amd import _hip
    windows = WINDOW_SETS["std3"]
    M_ = np.zeros((4, 10, 6))
    with pytest.raises(_hip.HipExtensionError):
        _hip.forward_host(M_, None, windows, device=[0, 99])
    with pytest.raises(_hip.HipExtensionError):
        _hip.forward_host(M_, None, windows, device=[0] * 5)        # at most 4 occurrences of one device
    y, st = _hip.forward_host(M_, None, windows, device=[0])         # and the library is usable afterwards
    assert not y.any() and not st.any()


def test_numpy_entry_points_first_then_torch_in_a_fresh_interpreter():
    """A program that starts with the numpy-only entry points (libmlpg_hip.so loaded before torch) and later uses the tensor API:
    one HIP runtime in the process, torch still sees the GPU, both calls give the oracle's result."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from nnmnkwii_amd import paramgen as G
assert "torch" not in sys.modules
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(0)
M, V = rng.randn(3, 50, 6), rng.rand(3, 50, 6) + 0.1
y = G.mlpg_batch(M, V, W3)                       # numpy in, numpy out: the host-pointer entry point
assert "torch" not in sys.modules
import torch
assert torch.cuda.is_available()
yt = G.mlpg_batch(torch.from_numpy(M).cuda(), torch.from_numpy(V).cuda(), W3).cpu().numpy()
from oracle import mlpg as O
yo, _, rc = O.mlpg_batch(M, V, W3)
assert rc == 0 and np.abs(y - yo).max() < 1e-10 and np.abs(yt - yo).max() < 1e-10
print("ok")
""" % (root, os.path.join(root, "tests", "golden"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-300:], out.stderr[-800:])


def test_every_visible_device_gets_its_share_of_the_chunks():
    """devices="all": the chunks of one call are dealt round-robin over every visible GPU (the library's per-device chunk
    counter, mlpg_hip_launch_count(100 + d)).  On a one-GPU box this checks the single device's count; on a node with
    several GPUs that each device ran ceil / floor of its share and that the result equals the single-device call's."""
    import torch
    from nnmnkwii_amd import _hip
    L = _hip.lib()
    nd = torch.cuda.device_count()
    windows = WINDOW_SETS["std3"]
    rng = np.random.RandomState(77)
    B, T, sd = 64, 500, 60                      # 69 MB of input: several chunks per device
    M_ = rng.randn(B, T, 3 * sd)
    V_ = rng.rand(B, T, 3 * sd) + 0.1
    entry, slot, first, count = _hip.host_chunk_plan(B, max(1, (64 << 20) // (2 * T * 3 * sd * 8)), nd)
    before = [int(L.mlpg_hip_launch_count(100 + d)) for d in range(nd)]
    y = _hip.forward_host(M_, V_, windows, device="all")[0]
    after = [int(L.mlpg_hip_launch_count(100 + d)) for d in range(nd)]
    got = [a - b for a, b in zip(after, before)]
    want = [int((entry == d).sum()) for d in range(nd)]
    assert got == want and sum(got) == len(entry) and min(got) >= 1
    y0 = _hip.forward_host(M_, V_, windows, device=0)[0]
    assert np.array_equal(y, y0)
    if nd > 1:
        # an explicit list in another order, and a device listed twice next to one listed once
        y2 = _hip.forward_host(M_, V_, windows, device=list(range(nd))[::-1])[0]
        y3 = _hip.forward_host(M_, V_, windows, device=[0, 1, 0])[0]
        assert np.array_equal(y2, y0) and np.array_equal(y3, y0)


def test_bench_under_the_drivers_launcher_with_the_rccl_backend_at_world_size_one():
    """`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 --gather`: the launch form the driver uses for the
    N > 1 legs, with the nccl (= RCCL) backend's init, barriers and all-gathers actually running on the GPU -- at world size 1,
    which is what a one-GPU box allows.  Plus tools/dbg/nccl_world1.py: the collectives sharding.py makes."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["NNMNKWII_BENCH_FORCE_DIST"] = "1"      # initialise the process group even at world size 1
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "32", "--frames", "300", "--precondition", "0", "--regions", "0", "--no-cpu-baseline", "--no-secondary",
                        "--no-traffic", "--gather"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["allgather_ms"] is not None and res["parity_rel_err_vs_oracle"] < 1e-9
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dbg", "nccl_world1.py")], capture_output=True, text=True, timeout=300,
                       env=dict(env, MASTER_PORT=str(port + 1 if port < 65000 else port - 1)), cwd=root)
    assert r.returncode == 0 and "nccl world-1 ok" in r.stdout, (r.stdout + r.stderr)[-2000:]
