#!/usr/bin/env bash
# oracle/build_reference.sh -- build the UNMODIFIED reference (Python + Cython)
# in a scratch directory OUTSIDE this repo so that tests/golden/make_golden.py
# can import it.  Nothing from /root/reference is copied into the repository;
# the scratch tree is throw-away.  Only works where /root/reference exists
# (the build container), not on the GPU box.
set -euo pipefail
REF=${1:-/root/reference}
DST=${2:-/tmp/oracle_ref}
if [ ! -d "$REF/nnmnkwii" ]; then echo "no reference at $REF" >&2; exit 3; fi
rm -rf "$DST" && mkdir -p "$DST"
cp -r "$REF/nnmnkwii" "$REF/setup.py" "$REF/README.md" "$DST/"
chmod -R u+w "$DST"
( cd "$DST" && python setup.py build_ext --inplace > build.log 2>&1 ) || { tail -20 "$DST/build.log"; exit 1; }
echo "__version__ = '0.1.3+oracle'" > "$DST/nnmnkwii/version.py"
echo "reference built at $DST (PYTHONPATH=$DST)"
