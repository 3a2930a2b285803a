#!/bin/bash
# quick A/B of env-tunable variants: prints ms/frame and per-kernel ms for each setting
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for pps in 3 4 6 8 12; do
  echo "== pairs_per_strip=$pps"
  FFTUP_LIBRARY=$GRAFT_REPO_ROOT/vkresample_amd/libfftup_knobs.so FFTUP_EXPERIMENT=pairs_per_strip=$pps python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'], d['kernel_ms'])"
done | tee $OUT/sweep.txt
