#!/usr/bin/env bash
# strip kernel A/B: each argument is a set of -D flags; rebuilds the forward f64 instantiation on the box and benches it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
  for rep in 1 2; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --regions 0 --no-check --algo 3 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('[$flags] kernel_ms %.4f steady %.4f GB/s %.1f frac %.3f' % (rf['kernel_ms'], rf['kernel_ms_steady'], rf['achieved'], rf['frac']))"
  done
done
