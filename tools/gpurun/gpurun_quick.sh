#!/usr/bin/env bash
# quick loop: wave-kernel tests + kernel timing
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_mlpg_gpu.py -m gpu -q -x 2>&1 | grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|assert " | head -20
python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); rf = r['roofline']
print('kernel_ms %.4f  frames/s %.3e  GB/s %.1f  frac %.3f' % (rf['kernel_ms'], r['value'], rf['achieved'], rf['frac']))"
