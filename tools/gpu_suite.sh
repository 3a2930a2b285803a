#!/bin/bash
# the whole GPU suite + smoke on the GPU box:  gpurun --timeout 2400 -- tools/gpu_suite.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-suite}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -s --timeout=900 ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest_gpu.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.txt | tail -15
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
