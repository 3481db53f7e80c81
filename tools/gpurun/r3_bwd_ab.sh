#!/usr/bin/env bash
# strip kernel backward A/B: each argument is a set of -D flags (backward instantiations rebuilt), parity check + timing per build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_bwd > /dev/null 2>&1
  echo "=== [$flags] $(timeout 300 python -m pytest tests/test_strip_gpu.py tests/test_parity_r2_gpu.py -x -q -k 'all_lengths or full_size or backward or grad or long_range' 2>&1 | tail -1)"
  timeout 300 python tools/bench_paths.py --only c2b 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('   %-22s %.4f ms  frac %.3f' % (r['path'], r['ms'], r['roofline_frac']))"
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_bwd > /dev/null 2>&1
