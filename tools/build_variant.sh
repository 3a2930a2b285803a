#!/bin/bash
# tools/build_variant.sh <tag> [extra hipcc flags]  ->  tools/scratch/lib_<tag>.so  (a library build for tools/gpu_ab.sh; test knobs compiled in)
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/scratch
python - "$tag" "$@" <<'PY'
import sys
import __graft_entry__ as g
print(g.build_variant("tools/scratch/lib_%s.so" % sys.argv[1], extra=sys.argv[2:]))
PY
