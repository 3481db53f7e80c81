#!/usr/bin/env bash
# Round 5, first GPU minutes: the verification debt of round 4 (full -m gpu suite incl. the shutdown test with fir_shutdown,
# tools/dbg/fir_soak.py), the LDS-shared FIR tiles (MLPG_FIR_SHARED=1: parity + A/B timing), the numpy -> numpy host path.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_first
timeout 1200 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1
echo "pytest rc=$?" >> ${O}_tests.log
tail -4 ${O}_tests.log
echo "== fir soak (shipped tiles)"
timeout 200 python tools/dbg/fir_soak.py 75 20260926 > ${O}_fir_soak.log 2>&1; echo "rc=$?" >> ${O}_fir_soak.log; tail -3 ${O}_fir_soak.log
echo "== fir shared: parity"
MLPG_FIR_SHARED=1 timeout 600 python -m pytest tests/test_fir_gpu.py tests/test_autograd_gpu.py -m gpu -x -q > ${O}_fir_shared_tests.log 2>&1
echo "rc=$?" >> ${O}_fir_shared_tests.log; tail -4 ${O}_fir_shared_tests.log
echo "== fir soak (shared tiles)"
MLPG_FIR_SHARED=1 timeout 200 python tools/dbg/fir_soak.py 60 7 > ${O}_fir_soak_shared.log 2>&1; echo "rc=$?" >> ${O}_fir_soak_shared.log; tail -3 ${O}_fir_soak_shared.log
for sw in 0 1; do
echo "== MLPG_FIR_SHARED=$sw"
MLPG_FIR_SHARED=$sw timeout 300 python - <<'PY' 2>&1 | tee -a ${O}_fir_ab.log
import sys, numpy as np, torch
sys.path.insert(0, ".")
from nnmnkwii_amd import _hip
W3 = [(0, 0, np.array([1.0])), (1, 1, np.array([-0.5, 0.0, 0.5])), (1, 1, np.array([1.0, -2.0, 1.0]))]
def timeit(fn, reps=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))
for (B, T, sd) in ((64, 500, 60), (256, 1000, 60)):
    m = torch.rand(B, T, 3 * sd, dtype=torch.float32, device="cuda")
    g = torch.randn(B, T, sd, dtype=torch.float32, device="cuda")
    tg = torch.rand(B, T, sd, dtype=torch.float32, device="cuda")
    f = timeit(lambda: _hip.forward(m, None, W3, None, algo=7, want_status=False))
    b = timeit(lambda: _hip.backward(None, g, W3, 3 * sd, out_dtype=torch.float32, algo=7, want_status=False))
    s = timeit(lambda: _hip.unit_mse_step(m, tg, W3))
    print("%d x %d x %d  forward %.4f ms  backward %.4f ms  step %.4f ms" % (B, T, sd, f, b, s))
PY
done
echo "== kernel durations of config 3, shared off / on"
for sw in 0 1; do
  MLPG_FIR_SHARED=$sw rocprofv3 --kernel-trace --stats -d ${O}_c3_sw$sw -o run -- python tools/bench_paths.py --only c3 > ${O}_c3_sw$sw.log 2>&1
  f=$(find ${O}_c3_sw$sw -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > ${O}_c3_sw$sw.txt 2>&1; rm -rf ${O}_c3_sw$sw
  grep -i "fir" ${O}_c3_sw$sw.txt | head -8
  grep '"path"' ${O}_c3_sw$sw.log | cut -c1-400
done
echo "== host path"
MLPG_HIP_HOST_TRACE=0 timeout 300 python tools/dbg/host_path_time.py 2>&1 | tee ${O}_host_path_time.txt
