#!/usr/bin/env bash
# Build a variant of the library HERE (no GPU needed) with extra -D flags for some translation units:
#   tools/dbg/build_variant.sh <tag> "<flags>" <source.hip> [<source.hip> ...]
# -> tools/dbg/bin/libmlpg_hip_<tag>.so (git-ignored; travels with the gpurun snapshot); select it with NNMNKWII_AMD_SO.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
tag=$1; flags=$2; shift 2
cd "$ROOT/nnmnkwii_amd/csrc"
mkdir -p "$ROOT/tools/dbg/bin" /tmp/variant_$tag
skip=""
objs=""
for src in "$@"; do
  o=/tmp/variant_$tag/${src%.hip}.o
  extra="-ffp-contract=fast"
  case $src in mlpg_fir.hip|mlpg_chunk_*.hip) extra="-ffp-contract=fast -mllvm -pragma-unroll-threshold=200000 -mllvm -unroll-threshold=200000";; dtw*|modspec*) extra="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function $extra $flags -c $src -o $o &
  skip="$skip ${src%.hip}.o"
  objs="$objs $o"
done
wait
rest=""
for o in *.o; do case " $skip " in *" $o "*) ;; *) rest="$rest $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/dbg/bin/libmlpg_hip_$tag.so" $rest $objs
echo "$ROOT/tools/dbg/bin/libmlpg_hip_$tag.so"
