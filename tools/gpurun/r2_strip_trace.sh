#!/usr/bin/env bash
# strip kernel item trace (start / all-arrived / level-3 done / end per item, 100 MHz ticks); $1 = extra -D flags
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MLPG_HIP_EXTRA_FLAGS="-DMLPG_STRIP_TRACE $1" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
MLPG_DUMP_STATUS=trace timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-check --algo 3 2> gpurun_out/trace.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('[$1] kernel_ms %.4f steady %.4f GB/s %.1f frac %.3f' % (rf['kernel_ms'], rf['kernel_ms_steady'], rf['achieved'], rf['frac']))"
grep -A12 "strip trace" gpurun_out/trace.err | cut -c1-900
