"""N > 1 path on CPU (-m "not gpu"): world_size-2 gloo processes exercise the batch sharding
and the result gather.  The per-shard compute is injected (the oracle, as the checker) because
the product has no CPU compute path; on GPUs the same code runs with the HIP kernels + RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partition():
    from nnmnkwii_amd.sharding import shard_range
    for n in (0, 1, 5, 8, 17, 256, 1024):
        for world in (1, 2, 3, 4, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            for (a, b), (c, d) in zip(parts, parts[1:]):
                assert b == c and b >= a
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cases import WINDOW_SETS
        from nnmnkwii_amd import sharding
        from oracle import dtw as OD
        from oracle import mlpg as O
        windows = WINDOW_SETS["std3"]
        rng = np.random.RandomState(0)          # same data on every rank (SPMD)
        T, sd = 40, 3
        M = rng.randn(B, T, 3 * sd)
        V = rng.rand(B, T, 3 * sd) + 0.1
        lengths = rng.randint(1, T + 1, size=B).astype(np.int32)

        def compute(m, v, w, L):
            return O.mlpg_batch(m, v, w, L)[0]

        full = sharding.mlpg_batch_sharded(M, V, windows, lengths, gather=True, compute=compute)
        ref = O.mlpg_batch(M, V, windows, lengths)[0]
        ok = full.shape == ref.shape and np.array_equal(full, ref)
        local, (lo, hi) = sharding.mlpg_batch_sharded(M, V, windows, lengths, gather=False, compute=compute)
        ok = ok and (lo, hi) == sharding.shard_range(B, rank, world) and np.array_equal(local, ref[lo:hi])
        # global variances + torch tensors in/out
        vg = rng.rand(3 * sd) + 0.1
        full_t = sharding.mlpg_batch_sharded(torch.from_numpy(M), torch.from_numpy(vg), windows, None, gather=True,
                                             compute=lambda m, v, w, L: torch.from_numpy(
                                                 O.mlpg_batch(m.numpy(), v.numpy(), w, L)[0]))
        ok = ok and torch.is_tensor(full_t) and np.array_equal(full_t.numpy(), O.mlpg_batch(M, vg, windows)[0])

        # DTW: shards of pairs, outputs padded to the global longest path
        X = np.zeros((B, 30, 4))
        Y = np.zeros((B, 36, 4))
        for n in range(B):
            a, b_ = rng.randint(10, 31), rng.randint(10, 37)
            X[n, :a] = np.cumsum(rng.randn(a, 4), 0)
            Y[n, :b_] = np.cumsum(rng.randn(b_, 4), 0)
        Xf, Yf = sharding.dtw_align_sharded(None, X, Y, transform=lambda xy: OD.dtw_align(xy[0], xy[1])[:2])
        Xo, Yo, _, _ = OD.dtw_align(X, Y)
        ok = ok and Xf.shape == Xo.shape and np.array_equal(Xf, Xo) and np.array_equal(Yf, Yo)

        # per-rank shards: every rank holds only its own utterances, of different counts (rank 1 may hold none)
        cut = B - 1 if B > 5 else B
        mine = slice(0, cut) if rank == 0 else slice(cut, B)
        full2 = sharding.mlpg_batch_sharded(M[mine], V[mine], windows, lengths[mine], gather=True, compute=compute,
                                            local_shards=True)
        ok = ok and full2.shape == ref.shape and np.array_equal(full2, ref)
        loc2, (lo2, hi2) = sharding.mlpg_batch_sharded(M[mine], V[mine], windows, lengths[mine], gather=False,
                                                       compute=compute, local_shards=True)
        ok = ok and (lo2, hi2) == (mine.start, mine.stop) and np.array_equal(loc2, ref[mine])
        Xf2, Yf2 = sharding.dtw_align_sharded(None, X[mine], Y[mine], transform=lambda xy: OD.dtw_align(xy[0], xy[1])[:2],
                                              local_shards=True)
        ok = ok and Xf2.shape == Xo.shape and np.array_equal(Xf2, Xo) and np.array_equal(Yf2, Yo)
        # ... and each rank padded its own shard to its OWN Tmax (what a per-process data loader does): the shards'
        # trailing shapes differ, the gather grows them to the global maximum first
        own_T = max(int(lengths[mine].max()), 1) if mine.stop > mine.start else 1
        full3 = sharding.mlpg_batch_sharded(M[mine, :own_T], V[mine, :own_T], windows, lengths[mine], gather=True,
                                            compute=compute, local_shards=True)
        t_glob = int(max(lengths[:cut].max(), lengths[cut:].max() if cut < B else 1))
        ok = ok and full3.shape == (B, t_glob, sd) and np.array_equal(full3, ref[:, :t_glob])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8])
def test_sharded_mlpg_and_dtw_gloo_world2(B):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` without a launcher starts 2 ranks itself (gloo dry run here: no GPU) and reports
    n_gpus = 2; without --dry-run-cpu it refuses to run on a node with fewer GPUs than asked for."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["dry_run"] is True and len(res["per_rank_ms_per_step"]) == 2
    assert res["scaling"] == "weak" and "in_run_single_gpu" in res
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                           capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_at_world_size_8_under_the_drivers_launcher():
    """The driver's own command form for the 8-GPU leg -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` -- as a gloo dry run: 8 ranks rendezvous, the barriers and
    the max-over-ranks reduction run, the per-rank `secondary` entries (configs 4 and 5) are reduced to the slowest rank and to
    whole-job rates, and rank 0 prints ONE line with n_gpus = 8."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--dry-run-cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 alone prints
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["dry_run"] is True and len(res["per_rank_ms_per_step"]) == 8
    assert res["scaling"] == "weak" and res["ms_per_step"] == max(res["per_rank_ms_per_step"])
    sec = res["secondary"]
    c4, c5 = sec["c4-fastdtw-kernel"], sec["c5-forward_streams-one-call"]
    assert c4["ms_slowest_rank"] == 8.0 and c5["ms_slowest_rank"] == 16.0          # rank r reported 1 + r / 2 (1 + r) ms
    assert abs(c4["pairs_per_s_whole_job"] - 8 * 128e3 / 8.0) < 1e-6               # 8 ranks' shares at the slowest rank's time
    assert abs(c5["frames_per_s_whole_job"] - 8 * 512 * 2000e3 / 16.0) < 1e-3
