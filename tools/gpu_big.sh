#!/bin/bash
# one-off: the FFTUP_BIG_TESTS variants the default suite skips, and a 400-case size-generic sweep:  gpurun -- tools/gpu_big.sh <tag>
TAG=${1:-big}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
FFTUP_BIG_TESTS=1 timeout 1500 python -m pytest tests/test_gpu_jit.py tests/test_gpu_sweep.py -m gpu -q --timeout=900 > $OUT/big.txt 2>&1; tail -3 $OUT/big.txt
FFTUP_SWEEP_N=400 FFTUP_SWEEP_SEED=777 timeout 1500 python -m pytest tests/test_gpu_sweep.py -m gpu -q --timeout=900 -k "test_sweep_against_oracle" > $OUT/sweep400.txt 2>&1; tail -3 $OUT/sweep400.txt
