#!/usr/bin/env python
"""bench.py -- MLPG frames/sec on BASELINE.json configs[1] (per GPU: 256 utterances
x T=1000 x 60-dim mgc static+delta+delta-delta = 180 columns, float64, per-frame
variances, standard 3 windows), weak-scaled over N GPUs (one process per GPU,
independent shards, no data-path collective).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (one mlpg_hip_forward launch through the C
ABI) over the rank's resident batch.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WINDOWS = [
    (0, 0, np.array([1.0])),
    (1, 1, np.array([-0.5, 0.0, 0.5])),
    (1, 1, np.array([1.0, -2.0, 1.0])),
]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--static-dim", type=int, default=60)
    ap.add_argument("--algo", type=int, default=0, help="0 auto, 1 generic, 2 wave")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-check", action="store_true", help="profiling/ablation only: skip status and parity checks")
    ap.add_argument("--gather", action="store_true", help="also time an RCCL all-gather of the outputs (reported separately)")
    return ap.parse_args()


_REF_WORKER = r"""
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import ref
G = ref.load()
n, T, D, seed = (int(a) for a in sys.argv[2:6])
W = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
rng = np.random.RandomState(seed)
m = rng.randn(T, D); v = rng.rand(T, D) + 0.1
G.mlpg(m, v, W)
sys.stdin.readline()              # start gun
t0 = time.perf_counter()
for _ in range(n):
    G.mlpg(m, v, W)
print(time.perf_counter() - t0)
"""


def _reference_pool(n_per_proc, T, D, cores, timeout):
    """The reference's mlpg on `cores` independent processes (it has no threading of its own).
    Plain subprocesses with a hard timeout: this leg must never be able to hang the bench."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", _REF_WORKER, ROOT, str(n_per_proc), str(T), str(D), str(100 + i)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
             for i in range(cores)]
    try:
        time.sleep(min(10.0, 1.0 + 0.05 * cores))          # let every process import and warm up
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        outs = [p.communicate(timeout=timeout)[0] for p in procs]
        wall = time.perf_counter() - t0
        busy = [float(o.strip().splitlines()[-1]) for o in outs]
        return cores * n_per_proc * T / max(max(busy), 1e-9), wall
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def cpu_baseline(T, D, seconds):
    """CPU path timed on the GPU box's host, bounded sample of the same workload.

    kind "reference": the reference's OWN numpy/bandmat/Cython code (oracle/_ref: its sources
    compiled unmodified by oracle/build_ref_so.sh), called the way the reference batches -- a
    Python loop of paramgen.mlpg over utterances (util/__init__.py:44-66) -- on ONE core (the
    reference is single-threaded).  kind "port": the C restatement oracle/mlpg_oracle.c, if the
    compiled reference is not present."""
    from oracle import mlpg as O
    from oracle import ref
    O.build()
    rng = np.random.RandomState(1234)
    if ref.available():
        G = ref.load()
        m = rng.randn(T, D)
        v = rng.rand(T, D) + 0.1
        y = G.mlpg(m, v, WINDOWS)
        assert np.array_equal(y, O.mlpg(m, v, WINDOWS)), "C oracle and compiled reference disagree"
        t0 = time.perf_counter()
        G.mlpg(m, v, WINDOWS)
        per = time.perf_counter() - t0
        n = int(max(8, min(4096, seconds / max(per, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(n):
            G.mlpg(m, v, WINDOWS)
        dt = time.perf_counter() - t0
        res = {
            "value": n * T / dt, "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": "%d utterances x T=%d x D=%d float64 through the reference's own paramgen.mlpg "
                      "(numpy + bandmat Cython, compiled unmodified into oracle/_ref), Python loop over utterances, "
                      "%.1f s on 1 core of %d" % (n, T, D, dt, os.cpu_count()),
        }
        # the same on many cores (independent processes; the reference itself has no threading)
        try:
            cores = max(1, min(os.cpu_count() or 1, 64))
            per_proc = max(2, int(0.2 * n))
            rate, wall = _reference_pool(per_proc, T, D, cores, timeout=max(60.0, 6 * seconds))
            res["pool"] = {"value": rate, "unit": "frames/s", "cores": cores,
                           "sample": "%d processes x %d utterances each, %.1f s wall" % (cores, per_proc, wall)}
        except Exception as e:  # the single-core figure is the reported baseline
            res["pool"] = {"error": str(e)[:200]}
        return res
    m = rng.randn(4, T, D)
    v = rng.rand(4, T, D) + 0.1
    t0 = time.perf_counter()
    O.mlpg_batch(m, v, WINDOWS)
    per4 = time.perf_counter() - t0
    n = int(max(8, min(4096, seconds / max(per4 / 4, 1e-6))))
    n -= n % 4
    reps = n // 4
    t0 = time.perf_counter()
    for _ in range(reps):
        O.mlpg_batch(m, v, WINDOWS)
    dt = time.perf_counter() - t0
    return {
        "value": n * T / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d utterances x T=%d x D=%d float64 through oracle/mlpg_oracle.c (C restatement of "
                  "paramgen.mlpg, bit-exact vs the reference), %.1f s on 1 core of %d" % (n, T, D, dt, os.cpu_count()),
    }


def measured_traffic(B, T, sd, algo):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (profiles/traffic.json),
    if it was taken on this exact workload; bench.py cannot run rocprofv3 on itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        w = t["workload"]
        if (w["batch_per_gpu"], w["frames"], w["static_dim"]) == (B, T, sd) and algo in (0, 2):
            return float(t["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from nnmnkwii_amd import _hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    B, T, sd = args.batch, args.frames, args.static_dim
    D = 3 * sd
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    means = torch.randn(B, T, D, dtype=torch.float64, device=dev, generator=gen)
    variances = torch.rand(B, T, D, dtype=torch.float64, device=dev, generator=gen) + 0.1

    def step():
        return _hip.forward(means, variances, WINDOWS, None, algo=args.algo, want_status=True)

    for _ in range(args.warmup):
        out, status = step()
    torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            dist.barrier(device_ids=[local_rank])

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        out, status = step()
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0

    # average launch duration from ONE pair of HIP events around the K back-to-back launches, recorded on the
    # stream the kernel is launched on (per-launch event pairs add ~20 us of command-processor time each)
    kern_ms = float(ev0.elapsed_time(ev1)) / args.steps
    if not args.no_check:
        assert int(status.abs().max().item()) == 0, "a system was not positive definite"
    elif os.environ.get("MLPG_DUMP_STATUS"):
        st = status.cpu().numpy()[:64 * 8].reshape(64, 8)
        print("phase cycles (setup, wait-tiles, lds->regs, dma-issue, assembly, solve, status, output), mean over 64 WGs:",
              [int(x) for x in st.mean(0)], "sum", int(st.mean(0).sum()), file=sys.stderr)

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    gather_ms = None
    if args.gather and world > 1:
        full = torch.empty((world,) + tuple(out.shape), dtype=out.dtype, device=dev)
        dist.all_gather_into_tensor(full, out)
        torch.cuda.synchronize(dev)
        barrier()
        g0 = time.perf_counter()
        dist.all_gather_into_tensor(full, out)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        # parity spot check outside the timed region (oracle = checker only)
        from oracle import mlpg as O
        O.build()
        yo = O.mlpg(means[0].cpu().numpy(), variances[0].cpu().numpy(), WINDOWS)
        err = float(np.abs(out[0].cpu().numpy() - yo).max() / np.abs(yo).max())
        if not args.no_check:
            assert err < 1e-9, err

        frames = world * B * T * args.steps
        alg_bytes = 56.0 * sd * B * T            # 56 B per (frame, static dim) per launch
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "MLPG frames/sec (batch, 60-dim mgc x3 windows)",
            "value": frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: %d utterances/GPU x T=%d x %d-dim mgc static+delta+delta-delta "
                            "(%d columns), float64, per-frame variances, std 3 windows; mlpg_hip_forward via C ABI"
                            % (B, T, sd, D),
                "batch_per_gpu": B, "frames": T, "static_dim": sd, "algo": args.algo,
                "parallelism": "batch-sharded x%d, no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(B, T, sd, args.algo),
                "traffic_unit": "bytes/launch (PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/traffic.json)",
                "kernel_ms": kern_ms,
                "algorithmic_bytes": alg_bytes,
            },
            "parity_rel_err_vs_oracle": err,
        }
        if gather_ms is not None:
            res["allgather_ms"] = gather_ms
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(T, D, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
