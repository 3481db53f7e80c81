#!/usr/bin/env bash
# oracle/build_ref_so.sh -- compile the reference's OWN sources for the MLPG path (where they lie
# under /root/reference) into extension modules under oracle/_ref/ (git-ignored, binaries only).
# No reference source is copied into the repository: Cython translates each .py/.pyx to C in a
# throw-away directory and gcc links the .so straight into oracle/_ref/.  Used (i) to validate
# the C restatement against the real thing on the GPU box and (ii) as bench.py's cpu_baseline of
# kind "reference".  Only runs where /root/reference exists; the GPU box uses the prebuilt files.
set -euo pipefail
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
[ -d "$REF/nnmnkwii" ] || { echo "no reference at $REF" >&2; exit 3; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
PYINC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python -c "import numpy; print(numpy.get_include())")
SUF=$(python -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
rm -rf "$OUT"; mkdir -p "$OUT"
# module list: <source relative to $REF>  (package __init__ modules are compiled too)
MODS="
nnmnkwii/paramgen/_bandmat/__init__.py
nnmnkwii/paramgen/_bandmat/core.pyx
nnmnkwii/paramgen/_bandmat/full.pyx
nnmnkwii/paramgen/_bandmat/tensor.pyx
nnmnkwii/paramgen/_bandmat/linalg.pyx
nnmnkwii/paramgen/_bandmat/misc.py
nnmnkwii/paramgen/mlpg_helper.pyx
nnmnkwii/paramgen/_mlpg.py
nnmnkwii/util/linalg.py
nnmnkwii/util/_linalg.pyx
"
for m in $MODS; do
  [ -f "$REF/$m" ] || { echo "skip missing $m"; continue; }
  base=${m%.*}
  c="$TMP/$(echo "$base" | tr '/' '_').c"
  python -m cython -3 -I "$REF" "$REF/$m" -o "$c" >/dev/null 2>"$TMP/cy.log" || { cat "$TMP/cy.log"; exit 1; }
  mkdir -p "$OUT/$(dirname "$m")"
  gcc -O2 -fPIC -shared -w -I"$PYINC" -I"$NPINC" "$c" -o "$OUT/$base$SUF"
done
echo "reference modules built under $OUT"
find "$OUT" -name "*.so" | sed "s|$OUT/||"
