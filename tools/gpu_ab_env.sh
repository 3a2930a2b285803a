#!/bin/bash
# A/B of library builds under a fixed environment, isolated kernel times only: ENVSTR="A=1 B=2" tools/gpu_ab_env.sh lib1.so lib2.so ...
export TMPDIR=/tmp
cp vkresample_amd/libfftup.so /tmp/libfftup_keep.so
for lib in "$@"; do
  cp $lib vkresample_amd/libfftup.so
  echo -n "$lib [$ENVSTR]: "
  env $ENVSTR python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/frame'%d['ms_per_frame'], {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done
cp /tmp/libfftup_keep.so vkresample_amd/libfftup.so
