#!/bin/bash
# build, then run a command on the GPU box:  tools/grun.sh <timeout seconds> '<command>'   (the box gets the tree as it is NOW,
# built libraries included: a stale libfftup.so measures the previous source)
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
