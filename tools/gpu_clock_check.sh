#!/bin/bash
# rocprofv3 --kernel-trace --stats ON fftup_profile_kernels / fftup_execute_ring of one plan (tools/clock_check.py): trace averages beside the events' figures
TAG=${1:-clock}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
run() {  # label, args
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cc_$$ -o cc -- python $R/tools/clock_check.py ${@:2} > /tmp/cc_$$.out 2>/dev/null)
  echo "== $1: ${@:2}"; grep "^{" /tmp/cc_$$.out
  python - <<PY
import csv, glob
f = glob.glob("/tmp/cc_$$/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "fftup" in r["Name"]:
        print("   trace: %-70s calls %5s  avg %8.2f us  min %8.2f  max %8.2f" % (r["Name"].split("(")[0].replace("void fftup::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf /tmp/cc_$$ /tmp/cc_$$.out
}
{
run "1080p, default plan (ring 8, 3 streams: ONE strip per compute unit), kernels one after the other" --mode profile
run "1080p, sequential plan (1 stream: TWO strips per compute unit for 256-thread fused kernels), kernels one after the other" --mode profile --streams 1
run "1080p, default plan, frames overlapping on 3 streams" --mode ring
run "2048x1024 fp32, default plan, kernels one after the other" --mode profile --width 2048 --height 1024
run "2048x1024 fp32, sequential plan" --mode profile --width 2048 --height 1024 --streams 1
run "2048x1024 fp32, default plan, frames overlapping on 3 streams" --mode ring --width 2048 --height 1024
} > $OUT/clock_check.txt 2>&1
cat $OUT/clock_check.txt
