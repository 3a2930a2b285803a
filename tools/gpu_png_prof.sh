#!/bin/bash
# rocprofv3 kernel stats of the device-side PNG path (tools/png_device_time.py):  tools/gpu_png_prof.sh <tag>  -> gpurun_out/<tag>/
TAG=${1:-png}; shift; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_png -o png -- python $R/tools/png_device_time.py --frames 32 "$@" > $OUT/rocprof_png.log 2>&1 < /dev/null)
f=$(find /tmp/prof_png -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/kernel_stats_png.csv; cut -d, -f1-4 "$f" | head -16; else echo "no kernel stats"; tail -5 $OUT/rocprof_png.log; fi
