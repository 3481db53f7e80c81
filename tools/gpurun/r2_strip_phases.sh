#!/usr/bin/env bash
# strip kernel: phase timers (s_memtime, wavefront 0) and ablations, each a rebuild of the forward f64 instantiation on the box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # flags label
  MLPG_HIP_EXTRA_FLAGS="$1" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
  MLPG_DUMP_STATUS=strip timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-check --algo 3 2> gpurun_out/phases_$2.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('$2: kernel_ms %.4f steady %.4f GB/s %.1f frac %.3f' % (rf['kernel_ms'], rf['kernel_ms_steady'], rf['achieved'], rf['frac']))"
  grep -E "strip phase|by strip|^    \[|start cycle" gpurun_out/phases_$2.err
}
run "-DMLPG_STRIP_TIMING" timing





run "-DMLPG_STRIP_ABLATE=1" nosync | head -1
run "" shipped | head -1
timeout 600 python -m pytest tests/test_strip_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -3
