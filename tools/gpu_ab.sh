#!/bin/bash
# A/B on the GPU box (one gpurun call = one box, so the comparison is fair):
#   [BENCH_ARGS="--preset config3"] [PROF=1] [REPS=2] tools/gpu_ab.sh variant...
# a variant is a library build (path ending in .so, relative to the repo root), a set of environment variables
# ("FFTUP_STREAMS=2 FFTUP_LIBRARY=.../libfftup_knobs.so FFTUP_EXPERIMENT=g_per_cu=2": the knobs need the test build of the library), or the word "base" (the in-tree library as it is).
# Prints ms/frame (overlapped, the bench's `value`) and the kernels' sequential durations (HIP events; PROF=1: rocprofv3 averages).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cp vkresample_amd/libfftup.so /tmp/libfftup_base.so
trap 'cp /tmp/libfftup_base.so $ROOT/vkresample_amd/libfftup.so' EXIT
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  envs=""
  case "$v" in
    base) cp /tmp/libfftup_base.so vkresample_amd/libfftup.so ;;
    *.so) cp "$v" vkresample_amd/libfftup.so ;;
    *) cp /tmp/libfftup_base.so vkresample_amd/libfftup.so; envs="$v" ;;
  esac
  echo -n "[$v] "
  if [ -n "$PROF" ]; then
    rm -rf /tmp/abprof
    (cd /tmp && env $envs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof -o ab -- python $ROOT/bench.py --steps 2 --warmup 1 --frames-per-step 32 --profile-iters 2 --repeats 1 --no-cpu-baseline --no-others $BENCH_ARGS > /dev/null 2>&1)
    python - <<'PY'
import csv, glob
f = glob.glob('/tmp/abprof/**/*kernel_stats.csv', recursive=True)[0]
print({r['Name'].split('(')[0].replace('void fftup::', '')[:48]: round(float(r['AverageNs']) / 1e3, 1)
       for r in csv.DictReader(open(f)) if 'fftup' in r['Name'] and 'unpack' not in r['Name']})
PY
  else
    env $envs python bench.py --steps 5 --warmup 1 --repeats 3 --no-cpu-baseline --no-others $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/frame'%d['ms_per_frame'], {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
  fi
done; done
