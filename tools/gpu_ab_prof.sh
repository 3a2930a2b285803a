#!/bin/bash
# like gpu_ab.sh but reports rocprofv3 kernel averages (ns) instead of event timings
export TMPDIR=/tmp
cp vkresample_amd/libfftup.so /tmp/libfftup_base.so
for lib in /tmp/libfftup_base.so "$@"; do
  cp $lib vkresample_amd/libfftup.so
  rm -rf /tmp/abprof; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abprof -o ab -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --frames-per-step 32 --profile-iters 2 --no-cpu-baseline > /dev/null 2>&1)
  echo "== $lib"
  python - <<'PY'
import csv,glob
f=glob.glob('/tmp/abprof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'fftup' in n and 'unpack' not in n: print('   %-40s %8.1f us' % (n.split('(')[0].replace('void fftup::','')[:40], float(r['AverageNs'])/1e3))
PY
done
cp /tmp/libfftup_base.so vkresample_amd/libfftup.so
