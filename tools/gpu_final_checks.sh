#!/bin/bash
# final checks of a build: the GPU suite, every by-default-skipped variant (FFTUP_BIG_TESTS), and a wide size sweep with a fresh seed
# usage: tools/gpu_final_checks.sh <tag> [seed] [n_generic] [n_jit]
export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/suite.txt
FFTUP_BIG_TESTS=1 python -m pytest tests/test_gpu_jit.py tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_bench.py::test_eight_rank_dry_run_of_config5 -m gpu -q 2>&1 | tail -4 > $OUT/big.txt
FFTUP_SWEEP_SEED=${2:-4242} FFTUP_SWEEP_N=${3:-600} FFTUP_SWEEP_JIT_N=${4:-60} python -m pytest tests/test_gpu_sweep.py -m gpu -q 2>&1 | tail -4 > $OUT/sweep.txt
for f in suite big sweep; do tail -n 1 $OUT/$f.txt; done
