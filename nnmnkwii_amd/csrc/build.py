"""Build nnmnkwii_amd/csrc/libmlpg_hip.so for gfx950 with hipcc (in-tree)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["capi.hip", "host_api.hip", "mlpg_generic.hip", "mlpg_wave.hip", "mlpg_wave_fwd_f64.hip", "mlpg_wave_fwd_f32.hip",
           "mlpg_wave_bwd_f64.hip", "mlpg_wave_bwd_f32.hip", "mlpg_wave_fused.hip", "mlpg_strip.hip", "mlpg_strip_fwd_f64.hip",
           "mlpg_strip_fwd_f32.hip", "mlpg_strip_bwd_f64.hip", "mlpg_strip_bwd_f32.hip", "mlpg_strip_multi_f64.hip", "mlpg_strip_multi_f32.hip", "mlpg_const.hip", "mlpg_const_fwd_f64.hip", "mlpg_const_fwd_f32.hip", "mlpg_const_bwd_f64.hip", "mlpg_const_bwd_f32.hip", "mlpg_const_multi_f64.hip", "mlpg_const_multi_f32.hip", "mlpg_chunk.hip", "mlpg_chunk_fwd_f64.hip", "mlpg_chunk_fwd_f32.hip", "mlpg_chunk_bwd_f64.hip", "mlpg_chunk_bwd_f32.hip", "mlpg_fir.hip", "dtw.hip", "dtw_fast.hip", "dtw_costs.hip", "modspec.hip", "modspec_dft.hip"]
HEADERS = ["common.h", "assemble.h", "mlpg_wave_impl.h", "mlpg_strip_impl.h", "mlpg_const_impl.h", "mlpg_chunk_impl.h", os.path.join("..", "..", "include", "mlpg_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# The DTW kernels must round exactly like the CPU oracle (separate multiply and add); the MLPG
# kernels are free to fuse multiply-adds.
# mlpg_chunk_*.hip: their per-chunk loop (24 frame slots, a few hundred instructions each) must be unrolled completely -- every array
# index a constant -- and is larger than the optimiser's default limit for "#pragma unroll" (left at the default the loop stays
# and the arrays land in scratch).
FILE_FLAGS = {"mlpg_fir.hip": ["-ffp-contract=fast", "-mllvm", "-pragma-unroll-threshold=200000", "-mllvm", "-unroll-threshold=200000"],
              **{"mlpg_chunk_%s_%s.hip" % (d, t): ["-ffp-contract=fast", "-mllvm", "-pragma-unroll-threshold=200000", "-mllvm", "-unroll-threshold=200000"]
                 for d in ("fwd", "bwd") for t in ("f64", "f32")},
              "dtw_fast.hip": ["-ffp-contract=off"], "dtw.hip": ["-ffp-contract=off"], "dtw_costs.hip": ["-ffp-contract=off"], "modspec.hip": ["-ffp-contract=off"], "modspec_dft.hip": ["-ffp-contract=off"]}
COST = {"mlpg_wave_bwd_f64.hip": 29, "mlpg_wave_bwd_f32.hip": 27, "mlpg_chunk_bwd_f64.hip": 26, "mlpg_chunk_bwd_f32.hip": 26, "mlpg_strip_bwd_f64.hip": 23,
        "mlpg_strip_bwd_f32.hip": 23, "mlpg_chunk_fwd_f64.hip": 20, "mlpg_chunk_fwd_f32.hip": 20, "mlpg_fir.hip": 18, "mlpg_wave_fwd_f64.hip": 18,
        "mlpg_wave_fwd_f32.hip": 15, "mlpg_generic.hip": 14, "mlpg_strip_fwd_f32.hip": 13, "mlpg_strip_fwd_f64.hip": 12, "mlpg_const_bwd_f32.hip": 12,
        "mlpg_const_bwd_f64.hip": 12, "mlpg_strip_multi_f64.hip": 9, "mlpg_wave_fused.hip": 9, "mlpg_strip_multi_f32.hip": 8}
EXTRA = [f for f in os.environ.get("MLPG_HIP_EXTRA_FLAGS", "").split() if f]
SO = os.path.join(HERE, "libmlpg_hip.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers_of(src):
    """Headers a source depends on (the two big kernel headers only matter to their own instantiation files)."""
    hs = [h for h in HEADERS if h not in ("mlpg_wave_impl.h", "mlpg_strip_impl.h", "mlpg_const_impl.h", "mlpg_chunk_impl.h")]
    if src.startswith("mlpg_wave_"):
        hs.append("mlpg_wave_impl.h")
    if src.startswith("mlpg_strip_"):
        hs.append("mlpg_strip_impl.h")
    if src.startswith("mlpg_const_"):
        hs.append("mlpg_const_impl.h")
    if src.startswith("mlpg_chunk"):
        hs.append("mlpg_chunk_impl.h")
    return [os.path.join(HERE, h) for h in hs] + [os.path.abspath(__file__)]


def build(force=False, verbose=False, only=None):
    """only: iterable of source-name prefixes to force-rebuild (kernel experiments with MLPG_HIP_EXTRA_FLAGS)."""
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        forced = force or (only is not None and any(src.startswith(pfx) for pfx in only))
        if forced or _stale(o, [s] + _headers_of(src)):
            jobs.append([hipcc, "-x", "hip", *FLAGS, *FILE_FLAGS.get(src, ["-ffp-contract=fast"]), *EXTRA, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        return r.stderr

    if jobs:
        # longest first (seconds of one hipcc run, measured in round 6): the pool finishes when its longest-running job does
        jobs.sort(key=lambda cmd: -COST.get(os.path.basename(cmd[-3]), 5))
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 8, 16)) as ex:
            for err in ex.map(run, jobs):
                if verbose and err.strip():
                    print(err)
    if force or jobs or _stale(SO, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs])
    return SO


if __name__ == "__main__":
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    print(build(force="--force" in sys.argv, verbose="--quiet" not in sys.argv, only=only or None))
