#!/bin/bash
# A/B alternative builds of the library: [BENCH_ARGS="--preset config3"] tools/gpu_ab.sh lib1.so lib2.so ...   (paths relative to repo root)
export TMPDIR=/tmp
cp vkresample_amd/libfftup.so /tmp/libfftup_base.so
for rep in 1 2; do
for lib in /tmp/libfftup_base.so "$@"; do
  cp $lib vkresample_amd/libfftup.so
  echo -n "$lib: "
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/frame'%d['ms_per_frame'], {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done; done
cp /tmp/libfftup_base.so vkresample_amd/libfftup.so
