#!/usr/bin/env bash
# strip kernel phase timers for several flag sets (each argument)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fl in "$@"; do
MLPG_HIP_EXTRA_FLAGS="-DMLPG_STRIP_TIMING $fl" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
MLPG_DUMP_STATUS=strip timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-check --algo 3 2> gpurun_out/phases3.err | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('[$fl] kernel_ms %.4f steady %.4f' % (rf['kernel_ms'], rf['kernel_ms_steady']))"
grep -E "strip phase" gpurun_out/phases3.err | sed 's/.*strips: //'
done
