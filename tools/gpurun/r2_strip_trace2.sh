#!/usr/bin/env bash
# strip kernel timeline (loads-done / arrived / level 3 / end per item) for each flag set
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fl in "$@"; do
MLPG_HIP_EXTRA_FLAGS="-DMLPG_STRIP_TRACE $fl" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_strip_fwd_f64 > /dev/null 2>&1
echo "== [$fl]"
MLPG_DUMP_STATUS=trace timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-check --algo 3 2>&1 >/dev/null | grep -A 12 "strip trace" | cut -c1-900
done
