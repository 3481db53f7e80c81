#!/usr/bin/env bash
# pipe kernel A/B: each argument is a set of -D flags; rebuilds the forward instantiations on the box, then runs the given script
# usage: r3_pipe_ab.sh <python script + args> -- "<flags>" ["<flags>" ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cmd=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do cmd+=("$1"); shift; done
shift
for flags in "$@"; do
  echo "=== [$flags]"
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_pipe_fwd > /dev/null 2>&1
  timeout 200 python "${cmd[@]}" 2>&1 | tail -12
done
