#!/usr/bin/env python
"""Secondary measurements (1 GPU) for the other BASELINE.json configs; one JSON line per path.

    python tools/bench_paths.py [--quick]

  lit : the LITERAL drop-in calls: one numpy -> numpy paramgen.mlpg per utterance at BASELINE config 1 and at one config-2
        utterance, the reference's loop shape over 256 utterances, autograd.mlpg on CPU tensors, DTWAligner.transform on one
        pair -- wall clock per call beside the reference's on the same host (litq: without the reference's 7 s mlpg_grad)
  c2h : config 2 end to end from host memory (numpy -> numpy), PCIe-inclusive wall clock
  c2g : config 2 with global (D,) variances and with unit variances (32 B per (frame, dim))
  c3  : autograd.unit_variance_mlpg forward+backward, B=64 x T=500 x 180, float32 tensors on the GPU
  c3m : autograd.mlpg (generic variances) forward+backward, one utterance T=500 x 180 float32
  c4  : DTWAligner on 128 pairs (1 GPU share of config 4), T in [700, 900], 25-dim, radius 1
  ms  : modspec_smoothing / modspec (n = 4096) of a config-2 sized trajectory batch 256 x 1000 x 60, float64
  c5  : Merlin-style acoustic paramgen mgc(60)+lf0(1)+bap(5), T=2000, B=512 (1 GPU share of config 5), float64;
        per-stream dense tensors, and the three streams in place from one (B, T, 198) batch (forward_streams)

Each line carries the GPU time (HIP events on the launch stream), the algorithmic bytes, GB/s, and a
bounded CPU baseline from the oracle on the same host (the checker, timed like bench.py's cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WINDOWS = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]


def gpu_time(fn, steps=10, warmup=2, settle_ms=None):
    """Median HIP-event time of one call.  After the warm-up calls the function is run for about `settle_ms` of device time
    (default 25, NNMNKWII_BENCH_SETTLE_MS; at most 400 calls) before the timed calls: the device's clocks take tens of milliseconds
    of work to settle and fall back within milliseconds of idling (profiles/r05_notes.md section 18), and every path here is
    measured behind a pause."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if settle_ms is None:
        settle_ms = float(os.environ.get("NNMNKWII_BENCH_SETTLE_MS", "25"))
    if settle_ms > 0:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        for _ in range(int(min(400, settle_ms / max(a.elapsed_time(b), 1e-3)))):
            fn()
    evs = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


def wall_time(fn, steps=1000, warmup=50, repeats=3):
    """Wall-clock milliseconds per call over a back-to-back loop with one synchronize at its end (best of `repeats`): what a
    training loop sees, and how the reference times its own loop (perf/autograd_mlpg_perf.py:56-86: time.time() around the loop).
    For host-bound steps the event pair of gpu_time() around every single call reads 1.5-2.5 x this (profiles/r06_notes.md section 10)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    return best


HBM_PEAK_GBS = 8000.0


_SINK = None      # run(sink=[...]) collects the lines instead of printing them (bench.py's `secondary` object)


def emit(**kw):
    if "GBps" in kw and kw["GBps"] is not None:
        kw["roofline_frac"] = kw["GBps"] / HBM_PEAK_GBS      # algorithmic bytes / time against the 8 TB/s HBM3E peak
    if _SINK is not None:
        _SINK.append(kw)
    else:
        print(json.dumps(kw), flush=True)


def fastdtw_window_cells(x, y, radius=1):
    """DP cells fastdtw visits for one pair, over all levels of the halving pyramid (the kernel's work measure beside
    bytes): the full matrix at the coarsest level, else the per-row intervals that upstream's __expand_window builds
    from the coarser path (union of (2r+1)^2 neighbourhoods, each coarse cell = 2x2 fine cells, one run per row)."""
    from oracle import dtw as OD
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if len(x) < radius + 2 or len(y) < radius + 2:
        return len(x) * len(y)
    xs = (x[0:len(x) - len(x) % 2:2] + x[1:len(x):2][: len(x) // 2]) / 2
    ys = (y[0:len(y) - len(y) % 2:2] + y[1:len(y):2][: len(y) // 2]) / 2
    cells = fastdtw_window_cells(xs, ys, radius)
    _, path = OD.fastdtw(xs, ys, radius)
    lo = np.full(len(xs) + 2 * radius + 2, 1 << 30)
    hi = np.full(len(xs) + 2 * radius + 2, -1)
    for i, j in path:
        for a in range(-radius, radius + 1):
            r = i + a + radius
            lo[r] = min(lo[r], j - radius)
            hi[r] = max(hi[r], j + radius)
    tot = 0
    for i in range(len(x)):
        r = i // 2 + radius
        if hi[r] < 0:
            continue
        tot += min(2 * hi[r] + 1, len(y) - 1) - max(2 * lo[r], 0) + 1
    return cells + tot


def _wall_us(fn, n, warm=5):
    """(median, min) wall-clock microseconds of one call of fn (host clock: these are host-memory calls)."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, float(np.min(ts)) * 1e6


def literal_calls(emit, quick=False, long_reference_backward=True):
    """The drop-in API exactly as `north_star` names it and as the reference's users call it: ONE numpy -> numpy
    `paramgen.mlpg(mean_frames (T, D), variance_frames, windows)` per utterance (paramgen/_mlpg.py:92), in a Python loop
    over utterances (util/__init__.py:56-66); `autograd.mlpg` on a CPU tensor; `DTWAligner.transform` on one pair.  Each
    beside the reference's own time on this host (oracle/_ref: the reference compiled unmodified; the checker, timed) and
    with the deviation from it.  Wall clock per call, everything included (staging, transfers, launch, synchronisation)."""
    import torch
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    from oracle import dtw as OD
    from oracle import mlpg as O
    from oracle import ref
    Gref = ref.load() if ref.available() else None
    ref_mlpg = Gref.mlpg if Gref is not None else O.mlpg
    ref_kind = "reference (oracle/_ref)" if Gref is not None else "port (oracle/mlpg_oracle.c)"
    rng = np.random.RandomState(1234)
    n_small = 50 if quick else 300

    def one(name, m, v, n):
        y = G.mlpg(m, v, WINDOWS)
        yr = ref_mlpg(m, v, WINDOWS)
        err = float(np.abs(y - yr).max() / max(np.abs(yr).max(), 1e-300))
        us, us_min = _wall_us(lambda: G.mlpg(m, v, WINDOWS), n)
        rus, rus_min = _wall_us(lambda: ref_mlpg(m, v, WINDOWS), max(10, n // 5), warm=2)
        emit(path=name, us_per_call=us, us_per_call_min=us_min, cpu_us_per_call=rus, cpu_us_per_call_min=rus_min, cpu_kind=ref_kind,
             speedup_vs_cpu=rus / us, T=int(m.shape[0]), D=int(m.shape[1]), frames_per_s=m.shape[0] / us * 1e6,
             rel_err_vs_cpu=err, out_dtype=str(y.dtype))

    # BASELINE config 1: T = 100, 2 static dims, 3 windows (means = rng.rand(100, 6), vars = rng.rand(6) tiled)
    m1 = rng.rand(100, 6)
    v1g = rng.rand(6)
    v1 = np.tile(v1g, (100, 1))
    one("lit-c1-paramgen.mlpg-T100-sd2", m1, v1, n_small)
    one("lit-c1-paramgen.mlpg-T100-sd2-global-variances", m1, v1g, n_small)
    # one utterance of BASELINE config 2: T = 1000, 60 static dims, per-frame variances
    m2 = rng.randn(1000, 180)
    v2 = rng.rand(1000, 180) + 0.1
    one("lit-c2utt-paramgen.mlpg-T1000-sd60", m2, v2, n_small)
    m2f, v2f = m2.astype(np.float32), v2.astype(np.float32)
    one("lit-c2utt-paramgen.mlpg-T1000-sd60-float32", m2f, v2f, n_small)
    # the reference's loop shape over config 2: 256 utterances, one call each (distinct arrays: 5.8 MB per utterance)
    nu = 64 if quick else 256
    utts = [(rng.randn(1000, 180), rng.rand(1000, 180) + 0.1) for _ in range(nu)]
    for m, v in utts[:8]:
        G.mlpg(m, v, WINDOWS)
    t0 = time.perf_counter()
    ys = [G.mlpg(m, v, WINDOWS) for m, v in utts]
    loop_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    yr = [ref_mlpg(m, v, WINDOWS) for m, v in utts[:32]]
    ref_s = (time.perf_counter() - t0) * nu / 32
    err = max(float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(ys, yr))
    t0 = time.perf_counter()
    yb = G.mlpg_batch(np.stack([m for m, _ in utts]), np.stack([v for _, v in utts]), WINDOWS)
    batch_s = time.perf_counter() - t0
    assert np.array_equal(yb[3], ys[3]) or float(np.abs(yb[3] - ys[3]).max()) < 1e-12
    emit(path="lit-c2-loop-per-utterance-calls", utterances=nu, ms=loop_s * 1e3, us_per_call=loop_s / nu * 1e6, frames_per_s=nu * 1000 / loop_s,
         cpu_ms=ref_s * 1e3, cpu_kind=ref_kind + ", 32 utterances timed, scaled", speedup_vs_cpu=ref_s / loop_s, rel_err_vs_cpu=err,
         ms_one_mlpg_batch_call_incl_stacking=batch_s * 1e3,
         note="[paramgen.mlpg(m, v, windows) for m, v in utterances]: the reference's own loop shape (util/__init__.py:56-66), numpy in, numpy out")
    del utts, ys, yr, yb
    # autograd.mlpg on CPU tensors (the reference's tensors are CPU tensors: autograd/_impl/mlpg.py:50-67), forward + backward.
    # One torch CPU thread, as the reference's CI pins it (OMP_NUM_THREADS=1, ci.yaml:16-17): with the default pool of one thread per
    # core (128 here) torch's own small CPU ops take milliseconds (a 60 k-element sum 27 us -> 2 us, an autograd step on (1000, 180) 2 ms -> 63 us,
    # tools/dbg/torch_cpu_threads.py) and its idle workers spin under the calls that follow; both sides of the comparison run under the same setting.
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    for name, T, sd, nrep in (("lit-c1-autograd.mlpg-cpu-tensor-T100-sd2", 100, 2, n_small), ("lit-c2utt-autograd.mlpg-cpu-tensor-T1000-sd60", 1000, 60, 30)):
        torch.manual_seed(1234)
        mt = torch.rand(T, 3 * sd, requires_grad=True)
        vt = torch.rand(T, 3 * sd) + 0.1

        def fb():
            mt.grad = None
            AF.mlpg(mt, vt, WINDOWS).sum().backward()

        us_f, _ = _wall_us(lambda: AF.mlpg(mt.detach(), vt, WINDOWS), nrep)
        us_fb, _ = _wall_us(fb, nrep)
        mn, vn = mt.detach().numpy(), vt.numpy()
        go_n = np.random.RandomState(5).randn(T, sd).astype(np.float32)
        us_g, _ = _wall_us(lambda: G.mlpg_grad(mn, vn, WINDOWS, go_n), nrep)    # the literal numpy -> numpy paramgen.mlpg_grad call

        class RefMLPG(torch.autograd.Function):
            # what the reference's node does (autograd/_impl/mlpg.py:50-67): paramgen.mlpg / mlpg_grad on .numpy() views
            @staticmethod
            def forward(ctx, means, variances):
                ctx.save_for_backward(means, variances)
                return torch.from_numpy(ref_mlpg(means.detach().numpy(), variances.detach().numpy(), WINDOWS))

            @staticmethod
            def backward(ctx, go):
                means, variances = ctx.saved_tensors
                return torch.from_numpy(Gref.mlpg_grad(means.detach().numpy(), variances.detach().numpy(), WINDOWS, go.numpy())), None

        rus_f, _ = _wall_us(lambda: RefMLPG.apply(mt.detach(), vt), 10, warm=1)
        rus_b = None
        if Gref is not None and (T <= 100 or (long_reference_backward and not quick)):
            go = np.ones((T, sd), dtype=np.float32)
            t0 = time.perf_counter()
            gr = Gref.mlpg_grad(mn, vn, WINDOWS, go)
            rus_b = (time.perf_counter() - t0) * 1e6
            fb()
            gerr = float(np.abs(mt.grad.numpy() - gr).max())
        else:
            gerr = None
        emit(path=name, us_forward=us_f, us_forward_backward=us_fb, us_paramgen_mlpg_grad=us_g, cpu_us_forward=rus_f, cpu_us_backward_mlpg_grad=rus_b,
             cpu_kind=ref_kind + " inside a torch.autograd.Function, as the reference's node calls it",
             grad_abs_err_vs_cpu=gerr, T=T, D=3 * sd, torch_cpu_threads=1)
    torch.set_num_threads(torch_threads)
    # paramgen.unit_variance_mlpg_matrix(windows, T) (a9: _mlpg.py:297-373; once per minibatch length in the reference's training loop)
    if not quick:
        for T in (100, 500):
            us_m, us_m_min = _wall_us(lambda: G.unit_variance_mlpg_matrix(WINDOWS, T), 10, warm=2)
            rus_m = merr = None
            if Gref is not None:
                t0 = time.perf_counter()
                Rr = Gref.unit_variance_mlpg_matrix(WINDOWS, T)
                rus_m = (time.perf_counter() - t0) * 1e6
                merr = float(np.abs(G.unit_variance_mlpg_matrix(WINDOWS, T) - Rr).max())
            emit(path="lit-unit_variance_mlpg_matrix-T%d" % T, us_per_call=us_m, us_per_call_min=us_m_min, cpu_us_per_call=rus_m, cpu_kind=ref_kind,
                 abs_err_vs_cpu=merr, speedup_vs_cpu=(rus_m / us_m if rus_m else None), T=T)
    # DTWAligner.transform on ONE pair (alignment.py:41-76 is a per-pair loop)
    a, b = 812, 777
    X = np.zeros((1, 900, 25))
    Y = np.zeros((1, 900, 25))
    X[0, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1
    Y[0, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
    al = DTWAligner()
    Xa, Ya = al.transform((X, Y))
    Xo, Yo = OD.dtw_align(X, Y, 1, use_c=True)[:2]
    same = bool(np.array_equal(Xa, Xo) and np.array_equal(Ya, Yo))
    us, us_min = _wall_us(lambda: al.transform((X, Y)), 50 if quick else 200)
    cus, _ = _wall_us(lambda: OD.dtw_align(X, Y, 1, use_c=True), 10, warm=1)
    emit(path="lit-dtw-DTWAligner.transform-one-pair", us_per_call=us, us_per_call_min=us_min, cpu_us_per_call=cus,
         cpu_kind="oracle/dtw_oracle.c through oracle/dtw.py dtw_align (C restatement of fastdtw; the reference's pure-Python fastdtw package is absent)",
         aligned_arrays_equal_oracle=same, Tx=a, Ty=b, D=25)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    run(only=args.only, quick=args.quick)


def run(only="", quick=False, sink=None, device_index=0):
    """Run the selected paths (comma-separated keys, empty = all); with `sink` (a list) the result dicts are appended to
    it instead of being printed."""
    global _SINK
    _SINK = sink
    try:
        _run(only, quick, device_index)
    finally:
        _SINK = None


def _run(only, quick, device_index):
    class _A(object):
        pass
    args = _A()
    args.only, args.quick = only, quick
    import torch
    from nnmnkwii_amd import _hip
    from nnmnkwii_amd import autograd as AF
    from nnmnkwii_amd import paramgen as G
    from nnmnkwii_amd.preprocessing.alignment import DTWAligner
    from oracle import dtw as OD
    from oracle import mlpg as O
    O.build()
    dev = torch.device("cuda", device_index)
    gen = torch.Generator(device=dev).manual_seed(1234)
    want = lambda k: not args.only or k in args.only.split(",")  # noqa: E731

    # ---- c2k: config 2 through every forward kernel (generic / wave-per-system / strip) ----
    if want("c2k"):
        B, T, sd = 256, 1000, 60
        m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        by = 56.0 * sd * B * T
        for name, algo in (("generic", 1), ("wave", 2), ("strip", 3)):
            ms = gpu_time(lambda: _hip.forward(m, v, WINDOWS, algo=algo, want_status=False), steps=20)
            emit(path="c2k-forward-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        for name, algo in (("wave", 2), ("strip", 3)):
            go = torch.randn(B, T, sd, dtype=torch.float64, device=dev, generator=gen)
            ms = gpu_time(lambda: _hip.backward(v, go, WINDOWS, 3 * sd, out_dtype=torch.float64, algo=algo, want_status=False))
            byb = 8.0 * 7 * sd * B * T
            emit(path="c2k-backward-f64-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=byb, GBps=byb / ms / 1e6)
        # The same shape with variances as TTS acoustic models have them: the dynamic features 100 x / 1000 x tighter
        # than the static ones.  The trajectory is then smooth over hundreds of frames, the strip kernel's 5-strip
        # level-3 window is rejected by its damping bound and every strip sweeps the whole utterance.
        vt = v.clone()
        vt[:, :, sd:2 * sd] *= 1e-2
        vt[:, :, 2 * sd:] *= 1e-3
        for name, algo in (("wave", 2), ("strip", 3)):
            ms = gpu_time(lambda: _hip.forward(m, vt, WINDOWS, algo=algo, want_status=False), steps=20)
            emit(path="c2t-forward-tight-dynamic-variances-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        vt = v.clone()                      # moderately tight (10 x / 100 x): the 9-strip window
        vt[:, :, sd:2 * sd] *= 1e-1
        vt[:, :, 2 * sd:] *= 1e-2
        ms = gpu_time(lambda: _hip.forward(m, vt, WINDOWS, algo=3, want_status=False), steps=20)
        emit(path="c2t-forward-moderately-tight-dynamic-variances-strip", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        del vt
        # delta_features (the step before MLPG): read sd, write 3 sd per frame
        x = torch.randn(B, T, sd, dtype=torch.float64, device=dev, generator=gen)
        ms = gpu_time(lambda: _hip.delta_features(x, WINDOWS))
        byd = 8.0 * 4 * sd * B * T
        emit(path="c2k-delta_features", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=byd, GBps=byd / ms / 1e6)
        # long utterances (T = 4000: beyond the wave kernel): strip vs generic
        B2, T2 = 64, 4000
        m2 = torch.randn(B2, T2, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        v2 = torch.rand(B2, T2, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        by2 = 56.0 * sd * B2 * T2
        for name, algo in (("generic", 1), ("strip", 3)):
            ms = gpu_time(lambda: _hip.forward(m2, v2, WINDOWS, algo=algo, want_status=False), steps=5)
            emit(path="long-T4000-forward-" + name, ms=ms, frames_per_s=B2 * T2 / ms * 1e3, alg_bytes=by2, GBps=by2 / ms / 1e6)
        v2[:, :, sd:2 * sd] *= 1e-2          # the same with tight dynamic variances: windows of 8 / 16 strips per side
        v2[:, :, 2 * sd:] *= 1e-3
        ms = gpu_time(lambda: _hip.forward(m2, v2, WINDOWS, algo=3, want_status=False), steps=5)
        emit(path="long-T4000-forward-tight-dynamic-variances-strip", ms=ms, frames_per_s=B2 * T2 / ms * 1e3, alg_bytes=by2, GBps=by2 / ms / 1e6)
        del m, v, x, m2, v2

    # ---- c2t: the config-2 shape with TIGHT dynamic variances through AUTO (the strip kernel's slower level-3 rungs) ----
    # Variances as acoustic models have them: the dynamic features 10 x - 1000 x tighter than the static ones.  The trajectory is
    # then smooth over hundreds of frames; the strip kernel's 3-strip level-3 window is rejected by its damping bound and the
    # strips climb the ladder (5 / 9 / 33 strips, the whole utterance).  bench.py's metric data (variances ~ U(0.1, 1.1) in all
    # three windows) never leaves the 3-strip window: these entries put the other rungs into the driver's record.
    if want("c2t"):
        B, T, sd = 256, 1000, 60
        m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        by = 56.0 * sd * B * T
        for name, s1, s2 in (("10x-100x", 1e-1, 1e-2), ("100x-1000x", 1e-2, 1e-3)):
            vt = v.clone()
            vt[:, :, sd:2 * sd] *= s1
            vt[:, :, 2 * sd:] *= s2
            ms = gpu_time(lambda: _hip.forward(m, vt, WINDOWS, want_status=False), steps=20)
            y, st = _hip.forward(m, vt, WINDOWS)
            yo = O.mlpg(m[B - 1].cpu().numpy(), vt[B - 1].cpu().numpy(), WINDOWS)
            err = float(np.abs(y[B - 1].cpu().numpy() - yo).max() / np.abs(yo).max())
            emit(path="c2t-forward-dynamic-variances-tighter-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6,
                 status_max=int(st.abs().max().item()), rel_err_vs_oracle_last_utterance=err)
            del vt
        del m, v

    # ---- c2g: global / unit variances ----
    if want("c2g"):
        B, T, sd = 256, 1000, 60
        m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        vg = torch.rand(3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        for name, var in (("global", vg), ("unit", None)):
            ms = gpu_time(lambda: _hip.forward(m, var, WINDOWS, want_status=False))
            by = 32.0 * sd * B * T
            emit(path="c2g-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        # the same variances, backward (constant-coefficient kernel): read grad_out, write 3 gradient rows per (frame, dim)
        go = torch.randn(B, T, sd, dtype=torch.float64, device=dev, generator=gen)
        for name, var in (("global", vg), ("unit", None)):
            ms = gpu_time(lambda: _hip.backward(var, go, WINDOWS, 3 * sd, out_dtype=torch.float64, want_status=False))
            by = 32.0 * sd * B * T
            emit(path="c2g-" + name + "-backward", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        del m, go
        # config 5's mgc stream (T = 2000, 512 utterances = one GPU's share) with global variances
        B5, T5 = 512, 2000
        m5 = torch.randn(B5, T5, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        ms = gpu_time(lambda: _hip.forward(m5, vg, WINDOWS, want_status=False))
        by = 32.0 * sd * B5 * T5
        emit(path="c5g-mgc-global-variances", ms=ms, frames_per_s=B5 * T5 / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        del m5
        # the reference's 5-tap test windows (tests/test_paramgen.py:21-26: P has half-bandwidth 4), config-2 shape, per-frame
        # variances: the chunked kernel (AUTO) and the natural-order kernel
        wide = [(0, 0, np.array([1.0])), (2, 2, np.array([1.0, -8.0, 0.0, 8.0, -1.0]) / 12.0),
                (2, 2, np.array([-1.0, 16.0, -30.0, 16.0, -1.0]) / 12.0)]
        mw_ = torch.randn(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        vw_ = torch.rand(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        byw = 56.0 * sd * B * T
        for name, algo in (("", 0), ("-natural-order-kernel", 1)):
            ms = gpu_time(lambda: _hip.forward(mw_, vw_, wide, algo=algo, want_status=False), steps=5 if algo else 10)
            emit(path="c2w-forward-5-tap-windows" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=byw, GBps=byw / ms / 1e6)
        del mw_, vw_
        # long utterances, per-frame variances (T = 4000: 63 strips per utterance, dealt to the XCD work lists in two blocks)
        B2, T2 = 64, 4000
        m2 = torch.randn(B2, T2, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
        v2 = torch.rand(B2, T2, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
        ms = gpu_time(lambda: _hip.forward(m2, v2, WINDOWS, want_status=False), steps=5)
        by2 = 56.0 * sd * B2 * T2
        emit(path="long-T4000-forward", ms=ms, frames_per_s=B2 * T2 / ms * 1e3, alg_bytes=by2, GBps=by2 / ms / 1e6)
        del m2, v2

    # ---- c2h: config 2 end to end from host memory (numpy in -> numpy out through paramgen.mlpg_batch): PCIe-inclusive ----
    if want("c2h"):
        B, T, sd = 256, 1000, 60
        rng = np.random.RandomState(1234)
        mh = rng.randn(B, T, 3 * sd)
        vh = rng.rand(B, T, 3 * sd) + 0.1
        G.mlpg_batch(mh, vh, WINDOWS)          # full-size warm-up: the library's staging buffers grow on first use
        by = 56.0 * sd * B * T

        def host_walls(m_, v_, nrep=6):
            """Wall clock per call, every call into FRESH output pages (the arrays are kept alive, so that neither a reused
            mapping nor the munmap of the previous 123 MB result falls into the timed region: round 4's entry timed
            `y = mlpg_batch(...)` in a loop, i.e. each call plus the release of the previous result -- 20.5 ms against the
            16.1 ms tools/dbg/host_path_time.py measured for the same call)."""
            keep, ts, ts_free = [], [], []
            for _ in range(nrep):
                t0 = time.perf_counter()
                keep.append(G.mlpg_batch(m_, v_, WINDOWS))
                ts.append(time.perf_counter() - t0)
            yprev = keep.pop()
            for _ in range(3):                 # the round-4 form: the previous result is released inside the timed region
                t0 = time.perf_counter()
                yprev = G.mlpg_batch(m_, v_, WINDOWS)
                ts_free.append(time.perf_counter() - t0)
            del keep, yprev
            return float(np.median(ts)), float(np.min(ts)), float(np.mean(ts_free))

        wall, wmin, wfree = host_walls(mh, vh)
        emit(path="c2h-numpy-to-numpy-mlpg_batch", ms=wall * 1e3, ms_min=wmin * 1e3, ms_with_release_of_previous_result=wfree * 1e3,
             frames_per_s=B * T / wall, alg_bytes=by,
             GBps=by / wall / 1e9, note="pageable host arrays through mlpg_hip_forward_host: chunked, staged by copy threads, transfers overlapped with the kernels; median of 6 calls into fresh output pages")
        # the same from pinned host arrays (nnmnkwii_amd._hip.pinned_empty): transferred in place, no staging copies
        mp, vp = _hip.pinned_empty(mh.shape), _hip.pinned_empty(vh.shape)
        mp[...] = mh
        vp[...] = vh
        G.mlpg_batch(mp, vp, WINDOWS)
        wall, wmin, wfree = host_walls(mp, vp)
        emit(path="c2h-numpy-to-numpy-mlpg_batch-pinned-inputs", ms=wall * 1e3, ms_min=wmin * 1e3, ms_with_release_of_previous_result=wfree * 1e3,
             frames_per_s=B * T / wall, alg_bytes=by,
             GBps=by / wall / 1e9, note="pinned inputs, pageable output: PCIe floor 12.9 ms (737 MB up at 57 GB/s, the 123 MB down overlapped); median of 6 calls into fresh output pages")
        yh = None
        del mh, vh, yh, mp, vp

    # ---- lit: the LITERAL drop-in calls -- one numpy -> numpy call per utterance, as a user of the reference writes them ----
    if want("lit") or want("litq"):
        literal_calls(emit, quick=args.quick, long_reference_backward=want("lit"))

    # ---- c2b: backward (mlpg_hip_backward) at config-2 scale, float64 and float32 ----
    if want("c2b"):
        B, T, sd = 256, 1000, 60
        for name, dt, esz in (("f64", torch.float64, 8), ("f32", torch.float32, 4)):
            v = torch.rand(B, T, 3 * sd, dtype=dt, device=dev, generator=gen) + 0.1
            go = torch.randn(B, T, sd, dtype=dt, device=dev, generator=gen)
            ms = gpu_time(lambda: _hip.backward(v, go, WINDOWS, 3 * sd, out_dtype=dt, want_status=False))
            by = float(esz) * (3 + 1 + 3) * sd * B * T     # read 3 variances + grad_out, write 3 grads per (frame, dim)
            emit(path="c2b-backward-" + name, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
            del v, go
        m32 = torch.randn(B, T, 3 * sd, dtype=torch.float32, device=dev, generator=gen)
        v32 = torch.rand(B, T, 3 * sd, dtype=torch.float32, device=dev, generator=gen) + 0.1
        ms = gpu_time(lambda: _hip.forward(m32, v32, WINDOWS, want_status=False))
        by = 4.0 * 7 * sd * B * T
        emit(path="c2-forward-f32", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
        del m32, v32

    # ---- c3: unit-variance autograd fwd+bwd ----
    if want("c3"):
        B, T, D = 64, 500, 180
        R = torch.from_numpy(G.unit_variance_mlpg_matrix(WINDOWS, T)).to(dev)
        means = torch.rand(B, T, D, device=dev, requires_grad=True)
        target = torch.rand(B, T, D // 3, device=dev)
        loss_fn = torch.nn.MSELoss()

        def step():
            means.grad = None
            y = AF.unit_variance_mlpg(R, means)
            loss_fn(y, target).backward()

        ms = gpu_time(step)
        nwall = 200 if args.quick else 1000
        ms_wall = wall_time(step, steps=nwall)

        def step_plain():      # the same loop with a trivial node in place of MLPG: what the framework itself costs per step
            means.grad = None
            loss_fn(means[..., :D // 3] * 1.0, target).backward()

        ms_wall_plain = wall_time(step_plain, steps=nwall)
        by = 1920.0 * B * T    # 960 B/frame forward + 960 B/frame backward
        # the same training step captured once into a HIP graph and replayed (eager mode is host-bound: a dozen
        # framework launches of a few microseconds each around two short kernels)
        ms_graph = None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            means.grad = None
            with torch.cuda.graph(graph):
                y_ = AF.unit_variance_mlpg(R, means)
                loss_ = loss_fn(y_, target)
                loss_.backward()
            ms_graph = gpu_time(graph.replay, steps=20)
        except Exception as e:  # noqa: BLE001
            ms_graph = "capture failed: %s" % str(e)[:120]
        # the two kernels alone (unit variances, float32), wave-per-system vs strip
        md = means.detach()
        god = torch.rand(B, T, D // 3, device=dev)
        for name, algo in (("wave", 2), ("strip", 3), ("fir", 7)):
            kf = gpu_time(lambda: _hip.forward(md, None, WINDOWS, algo=algo, want_status=False), steps=20)
            kb = gpu_time(lambda: _hip.backward(None, god, WINDOWS, D, out_dtype=torch.float32, algo=algo, want_status=False), steps=20)
            emit(path="c3-kernels-" + name, ms_forward=kf, ms_backward=kb, ms=kf + kb, alg_bytes=by, GBps=by / (kf + kb) / 1e6)
        # reference CPU form: dense R @ means on torch CPU (autograd/_impl/mlpg.py:138,158), 1 thread
        torch.set_num_threads(1)
        Rc, mc = R.cpu(), means.detach().cpu().requires_grad_()
        tc = target.cpu()
        t0 = time.perf_counter()
        nrep = 1 if args.quick else 3
        for _ in range(nrep):
            mc.grad = None
            rm = mc.view(B, T, 3, -1).transpose(1, 2).contiguous().view(B, -1, D // 3)
            loss_fn(torch.matmul(Rc, rm), tc).backward()
        cpu_s = (time.perf_counter() - t0) / nrep
        # the same step as ONE fused launch (autograd.unit_variance_mlpg_mse_loss -> mlpg_hip_unit_mse_step)
        def step_fused():
            means.grad = None
            AF.unit_variance_mlpg_mse_loss(R, means, target).backward()

        ms_fused = gpu_time(step_fused, steps=20)
        ms_fused_wall = wall_time(step_fused, steps=nwall)
        ms_fused_kernel = gpu_time(lambda: _hip.unit_mse_step(md, target, WINDOWS), steps=20)
        full = torch.full((B,), T, dtype=torch.int32, device=dev)     # (a lengths vector keeps the call on the one-launch kernel)
        ms_one_launch = gpu_time(lambda: _hip.unit_mse_step(md, target, WINDOWS, lengths=full), steps=20)
        ms_fused_graph = None
        try:
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            means.grad = None
            side2 = torch.cuda.Stream()
            side2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side2):
                for _ in range(3):
                    step_fused()
            torch.cuda.current_stream().wait_stream(side2)
            torch.cuda.synchronize()
            means.grad = None
            with torch.cuda.graph(g2, stream=side2):   # (the stream that was warmed up: the step's workspace is per stream and is
                AF.unit_variance_mlpg_mse_loss(R, means, target).backward()     # never created inside a capture, _hip._mse_workspace)
            ms_fused_graph = gpu_time(g2.replay, steps=20)
        except Exception as e:  # noqa: BLE001
            ms_fused_graph = "capture failed: %s" % str(e)[:120]
        emit(path="c3-fused-unit-mse-step", ms=ms_fused_kernel, ms_one_launch_kernel=ms_one_launch, ms_autograd_eager=ms_fused,
             ms_autograd_eager_wall_loop=ms_fused_wall, ms_hip_graph_replay=ms_fused_graph,
             frames_per_s=B * T / ms_fused_kernel * 1e3, alg_bytes=by + 4.0 * 60 * B * T, GBps=(by + 4.0 * 60 * B * T) / ms_fused_kernel / 1e6,
             note="ms = one mlpg_hip_unit_mse_step call (float32 without lengths: the FIR form, two launches, nothing allocated; "
                  "ms_one_launch_kernel: the wave-per-system kernel that takes the call when lengths are given): forward + MSE loss + backward of config 3; "
                  "ms_autograd_eager / ms_hip_graph_replay = the same through autograd.unit_variance_mlpg_mse_loss(...).backward(); "
                  "ms_autograd_eager_wall_loop: wall clock per step of %d back-to-back eager steps, one synchronize at the end" % nwall)
        emit(path="c3-unit-variance-autograd-fwd+bwd", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by,
             GBps=by / ms / 1e6, ms_wall_loop=ms_wall, ms_wall_loop_plain_torch_ops_no_mlpg_node=ms_wall_plain, ms_hip_graph_replay=ms_graph,
             note="ms: median HIP-event pair around ONE eager step (host-bound: the pair reads the host's issue time of a step that starts on an idle "
                  "device); ms_wall_loop: wall clock per step of %d back-to-back eager steps with one synchronize at the end -- how the reference "
                  "times this loop (perf/autograd_mlpg_perf.py:56-86); ..._plain_torch_ops_no_mlpg_node: the same loop with `means[..., :60] * 1.0` "
                  "in place of the MLPG node" % nwall,
             GBps_hip_graph_replay=(by / ms_graph / 1e6 if isinstance(ms_graph, float) else None),
             cpu_dense_matmul_1thread_frames_per_s=B * T / cpu_s)

    # ---- c3m: generic autograd.mlpg fwd+bwd (one utterance, like the reference's MLPG) ----
    if want("c3m"):
        T, D = 500, 180
        m1 = torch.rand(T, D, device=dev, requires_grad=True)
        v1 = torch.rand(T, D, device=dev) + 0.1
        tg = torch.rand(T, D // 3, device=dev)
        AF._mlpg.CHECK_STATUS = False

        def step1():
            m1.grad = None
            torch.nn.MSELoss()(AF.mlpg(m1, v1, WINDOWS), tg).backward()

        ms = gpu_time(step1)
        AF._mlpg.CHECK_STATUS = True
        t0 = time.perf_counter()
        O.mlpg(m1.detach().cpu().numpy(), v1.cpu().numpy(), WINDOWS)
        cpu_fwd = time.perf_counter() - t0
        emit(path="c3m-autograd.mlpg-fwd+bwd-1utt", ms=ms, frames_per_s=T / ms * 1e3,
             cpu_oracle_forward_only_frames_per_s=T / cpu_fwd,
             note="reference backward is O(T^2) dense solve_banded: 12.7 ms per static dim at T=500 (SURVEY 6)")

    # ---- c4: DTW (c4q: the kernel on one GPU's share only -- the per-rank leg of bench.py --gpus N) ----
    if want("c4") or want("c4q"):
        share_only = want("c4q") and not want("c4")
        N = 32 if args.quick else 128
        rng = np.random.RandomState(1234)
        Tx = Ty = 900
        X = np.zeros((N, Tx, 25))
        Y = np.zeros((N, Ty, 25))
        for n in range(N):
            a, b = rng.randint(700, 901, size=2)
            X[n, :a] = np.cumsum(rng.randn(a, 25), 0) * 0.1
            Y[n, :b] = np.cumsum(rng.randn(b, 25), 0) * 0.1
        Xd, Yd = torch.from_numpy(X).to(dev), torch.from_numpy(Y).to(dev)
        lenx, leny = _hip.trim_lengths(Xd), _hip.trim_lengths(Yd)
        ms = gpu_time(lambda: _hip.fastdtw_l2(Xd, Yd, lenx, leny, 1))
        pi, pj, pl, cost = _hip.fastdtw_l2(Xd, Yd, lenx, leny, 1)
        plen = pl.cpu().numpy()
        by = float(((lenx + leny).sum().item()) * 25 * 8 + 8 * plen.sum())
        ncpu = 2 if share_only else (8 if args.quick else 32)
        t0 = time.perf_counter()
        for n in range(ncpu):
            OD.fastdtw(X[n, :int(lenx[n])], Y[n, :int(leny[n])], 1)
        cpu_s = (time.perf_counter() - t0) / ncpu
        ms_full = None if share_only else gpu_time(lambda: DTWAligner().transform((X, Y)), steps=8, warmup=3)
        # DP cells over all pyramid levels (mean of 8 pairs): the kernel is latency-bound, cells/s is its work rate
        cells_per_pair = float(np.mean([fastdtw_window_cells(X[n, :int(lenx[n])], Y[n, :int(leny[n])], 1) for n in range(8)]))
        if not args.quick and not share_only:
            # all 1024 pairs of config 4 on one GPU (the pairs repeated 8 times)
            X8, Y8 = Xd.repeat(8, 1, 1).contiguous(), Yd.repeat(8, 1, 1).contiguous()
            lx8, ly8 = lenx.repeat(8).contiguous(), leny.repeat(8).contiguous()
            ms8 = gpu_time(lambda: _hip.fastdtw_l2(X8, Y8, lx8, ly8, 1), steps=5)
            emit(path="c4-fastdtw-kernel-1024pairs", pairs=8 * N, ms=ms8, pairs_per_s=8 * N / ms8 * 1e3, alg_bytes=8 * by,
                 GBps=8 * by / ms8 / 1e6, dp_cells_per_pair=cells_per_pair, dp_cells_per_s=cells_per_pair * 8 * N / ms8 * 1e3)
            del X8, Y8
            # the same 1024 pairs from HOST memory through mlpg_hip_fastdtw_host (device-side trim, chunks of pairs
            # on two streams, PCIe-inclusive wall clock), pageable and pinned inputs
            host_legs = (not args.only) or "c4h" in args.only.split(",")
            Xh, Yh = (np.tile(X, (8, 1, 1)), np.tile(Y, (8, 1, 1))) if host_legs else (None, None)
            for nm, (a_, b_) in () if not host_legs else (("pageable", (Xh, Yh)), ("pinned", (_hip.pinned_empty(Xh.shape), _hip.pinned_empty(Yh.shape)))):
                if nm == "pinned":
                    a_[...] = Xh
                    b_[...] = Yh
                _hip.fastdtw_host(a_, b_, 1)
                t0 = time.perf_counter()
                for _ in range(3):
                    _hip.fastdtw_host(a_, b_, 1)
                wall = (time.perf_counter() - t0) / 3 * 1e3
                emit(path="c4h-fastdtw-host-1024pairs-" + nm, pairs=8 * N, ms=wall, pairs_per_s=8 * N / wall * 1e3,
                     host_bytes=float(Xh.nbytes + Yh.nbytes))
        emit(path="c4-fastdtw-kernel", pairs=N, ms=ms, pairs_per_s=N / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6,
             dp_cells_per_pair=cells_per_pair, dp_cells_per_s=cells_per_pair * N / ms * 1e3,
             cpu_oracle_c_pairs_per_s=1.0 / cpu_s, transform_numpy_to_numpy_ms=ms_full,
             note="paths bit-exact vs the oracle's restatement of third-party fastdtw (parity unpinned: the package is absent)")

    # ---- ms: modulation-spectrum smoothing of the config-2 output batch (the step after MLPG) ----
    if want("ms"):
        B, T, sd, n = 256, 1000, 60, 4096
        x = torch.randn(B, T, sd, dtype=torch.float64, device=dev, generator=gen)
        ms = gpu_time(lambda: _hip.modspec_smoothing(x, n, 500, True), steps=5)
        by = 16.0 * B * T * sd                      # read + write the trajectory once
        # FFT arithmetic: 2 transforms x 5 n log2 n flops per column
        fl = 2 * 5.0 * n * 12 * B * sd
        xc = x[:4].cpu().numpy()
        t0 = time.perf_counter()
        for b_ in range(4):
            s_ = np.fft.rfft(xc[b_], n=n, axis=0)
            a_ = np.abs(s_)
            a_[500:] = 1.0
            np.fft.irfft(a_ * np.exp(1j * np.angle(s_)), n=n, axis=0)[:T]
        cpu_s = (time.perf_counter() - t0) / 4
        emit(path="ms-modspec_smoothing-n4096", ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6,
             fft_GFLOPs=fl / ms / 1e6, cpu_numpy_fft_frames_per_s=T / cpu_s)
        ms2 = gpu_time(lambda: _hip.modspec(x, n), steps=5)
        emit(path="ms-modspec-n4096", ms=ms2, frames_per_s=B * T / ms2 * 1e3)
        # a DFT length outside the in-LDS FFT: the direct transform (modspec_dft.hip)
        nd = 5000
        ms3 = gpu_time(lambda: _hip.modspec_smoothing(x, nd, 600, True), steps=3, warmup=1)
        t0 = time.perf_counter()
        for b_ in range(2):
            s_ = np.fft.rfft(xc[b_], n=nd, axis=0)
            np.fft.irfft(np.abs(s_) * np.exp(1j * np.angle(s_)), n=nd, axis=0)[:T]
        cpu3 = (time.perf_counter() - t0) / 2
        emit(path="ms-modspec_smoothing-n5000-direct", ms=ms3, frames_per_s=B * T / ms3 * 1e3, alg_bytes=by, GBps=by / ms3 / 1e6,
             cpu_numpy_fft_frames_per_s=T / cpu3)
        del x

    # ---- c5: Merlin-style multi-stream (c5q: the one-call form only) ----
    if want("c5") or want("c5q"):
        B, T = (128 if args.quick else 512), 2000
        tot_ms, tot_by = 0.0, sum(56.0 * sd_ * B * T for sd_ in (60, 1, 5))
        tot_by0, tot_by = tot_by, 0.0
        for name, sd in (("mgc", 60), ("lf0", 1), ("bap", 5)) if want("c5") else ():
            m = torch.randn(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen)
            v = torch.rand(B, T, 3 * sd, dtype=torch.float64, device=dev, generator=gen) + 0.1
            ms = gpu_time(lambda: _hip.forward(m, v, WINDOWS, want_status=False), steps=5)
            by = 56.0 * sd * B * T
            emit(path="c5-" + name, batch=B, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by, GBps=by / ms / 1e6)
            tot_ms += ms
            tot_by += by
            del m, v
        if want("c5"):
            emit(path="c5-all-streams", batch=B, ms=tot_ms, frames_per_s=B * T / tot_ms * 1e3, alg_bytes=tot_by,
                 GBps=tot_by / tot_ms / 1e6)
        tot_by = tot_by0
        # the same three streams consumed in place from ONE (B, T, 198) batch: mlpg_hip_forward_streams
        m = torch.randn(B, T, 198, dtype=torch.float64, device=dev, generator=gen)
        v = torch.rand(B, T, 198, dtype=torch.float64, device=dev, generator=gen) + 0.1
        streams = [(0, 60, WINDOWS), (180, 1, WINDOWS), (183, 5, WINDOWS)]
        ms = gpu_time(lambda: _hip.forward_streams(m, v, streams, want_status=False), steps=5)
        emit(path="c5-forward_streams-one-call", batch=B, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=tot_by,
             GBps=tot_by / ms / 1e6)
        del v
        # the same call with GLOBAL (D,) variances -- what a Merlin-style pipeline passes (util/files.py:90-115: one variance
        # vector per stream for the whole corpus): 32 B per (frame, dim); the streams merged on the constant-coefficient kernel
        vg5 = torch.rand(198, dtype=torch.float64, device=dev, generator=gen) + 0.1
        by_g = 32.0 * 66 * B * T
        ms = gpu_time(lambda: _hip.forward_streams(m, vg5, streams, want_status=False), steps=5)
        n0 = [int(_hip.lib().mlpg_hip_launch_count(k)) for k in range(9)]
        _hip.forward_streams(m, vg5, streams, want_status=False)
        n1 = [int(_hip.lib().mlpg_hip_launch_count(k)) for k in range(9)]
        emit(path="c5g-forward_streams-one-call-global-variances", batch=B, ms=ms, frames_per_s=B * T / ms * 1e3, alg_bytes=by_g,
             GBps=by_g / ms / 1e6, launches={k: b - a for k, (a, b) in enumerate(zip(n0, n1)) if b != a},
             note="launches: kernel family -> launches of one call (8 = merged constant-coefficient, 1 = wave-per-system for the 2 dims left over)")
        ms_sep = 0.0
        for in_col, sd_, _w in streams:       # the three streams as three calls on dense tensors, for comparison
            md = m[:, :, in_col:in_col + 3 * sd_].contiguous()
            vd = vg5[in_col:in_col + 3 * sd_].contiguous()
            ms_sep += gpu_time(lambda: _hip.forward(md, vd, WINDOWS, want_status=False), steps=5)
            del md
        emit(path="c5g-three-calls-global-variances", batch=B, ms=ms_sep, frames_per_s=B * T / ms_sep * 1e3, alg_bytes=by_g, GBps=by_g / ms_sep / 1e6,
             note="dense copies of the three streams' columns, one mlpg_hip_forward each")
        # ... and as three calls IN PLACE (one stream per forward_streams call: nothing to merge) -- what the one-call form competes with
        ms_inplace = sum(gpu_time(lambda: _hip.forward_streams(m, vg5, [st_]), steps=5) for st_ in streams)
        emit(path="c5g-three-calls-in-place-global-variances", batch=B, ms=ms_inplace, frames_per_s=B * T / ms_inplace * 1e3, alg_bytes=by_g,
             GBps=by_g / ms_inplace / 1e6)
        del m


if __name__ == "__main__":
    main()
