#!/bin/bash
# A/B of environment settings on the current library: tools/gpu_env_ab.sh "VAR=a" "VAR=b" ... [-- bench args]
export TMPDIR=/tmp
sets=(); args=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; args=("$@"); break; fi; sets+=("$1"); shift; done
for rep in 1 2; do
for s in "${sets[@]}"; do
  echo -n "$s: "
  env $s python bench.py --steps 5 --warmup 1 --no-cpu-baseline "${args[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/frame'%d['ms_per_frame'], {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done; done
