#!/bin/bash
# one-off larger sweep of the plan-time specialised plans incl. the fused 8-bit store:  gpurun -- tools/gpu_sweep_u8.sh <tag> [n]
TAG=${1:-sweep_u8}; N=${2:-150}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
FFTUP_SWEEP_JIT_N=$N FFTUP_SWEEP_SEED=4242 timeout 3000 python -m pytest tests/test_gpu_sweep.py -m gpu -q --timeout=900 -k specialised > $OUT/sweep.txt 2>&1
tail -5 $OUT/sweep.txt
