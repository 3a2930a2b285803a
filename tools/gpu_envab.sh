#!/bin/bash
# A/B of environment-variable variants: tools/gpu_envab.sh "VAR=1" "VAR=2 OTHER=3" ...
export TMPDIR=/tmp
for rep in 1 2; do
for envs in "$@"; do
  echo -n "[$envs] "
  env $envs python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/frame'%d['ms_per_frame'], {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done; done
