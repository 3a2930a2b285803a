#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, bench line, rocprofv3 kernel stats.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > $OUT/rocminfo.txt 2>&1
nproc >> $OUT/rocminfo.txt
timeout 900 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest_gpu.txt 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.txt | tail -15
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/prof -name "*kernel_stats*" | head; 
for f in $(find $OUT/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
