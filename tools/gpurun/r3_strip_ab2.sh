#!/usr/bin/env bash
# strip kernel A/B with a correctness check: each argument is a set of -D flags (forward f64 and f32 instantiations rebuilt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd > /dev/null 2>&1
  echo "=== [$flags] $(timeout 300 python -m pytest tests/test_strip_gpu.py -x -q -k 'all_lengths or full_size or tight or long_range or not_pd' 2>&1 | tail -1)"
  for rep in 1 2; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --regions 0 --algo 3 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('   kernel_ms %.4f steady %.4f frac %.3f' % (rf['kernel_ms'], rf['kernel_ms_steady'], rf['frac']))"
  done
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd > /dev/null 2>&1
