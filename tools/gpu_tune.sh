#!/bin/bash
# what the plan-time tuner decides and what it buys: tools/gpu_tune.sh <tag>
TAG=${1:-tune}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export FFTUP_CACHE_DIR=/tmp/fftup_tune_cache; rm -rf $FFTUP_CACHE_DIR
{
for cfg in "640 480 2" "720 480 2" "800 600 2" "1000 1000 2" "1440 900 2" "1600 900 2" "1920 1200 2" "2000 2000 2" "2560 1440 2" "3840 2160 2" "1280 720 1.5" "1920 1080 1.5" "2560 1440 1.5" "1920 1080 3" "1280 720 3"; do
  set -- $cfg
  for mode in default tuned; do
    if [ $mode = tuned ]; then export FFTUP_LIBRARY=$GRAFT_REPO_ROOT/vkresample_amd/libfftup_knobs.so FFTUP_EXPERIMENT=jit_tune=1 FFTUP_JIT_VERBOSE=1; else unset FFTUP_EXPERIMENT FFTUP_JIT_VERBOSE; rm -f $FFTUP_CACHE_DIR/wisdom.txt; fi
    python bench.py --width $1 --height $2 --upscale $3 --no-cpu-baseline --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 > $OUT/b_$1x$2_u$3_$mode.json 2> $OUT/err_$1x$2_u$3_$mode.txt
    grep "fftup: tuning" $OUT/err_$1x$2_u$3_$mode.txt | sed 's/^/    /'
    python - $OUT/b_$1x$2_u$3_$mode.json $mode <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-40s %-8s %8.1f us/frame  frac %.3f  kernels(us) %s" % (d["config"]["workload"].split(",")[0], sys.argv[2], d["ms_per_frame"]*1e3, d["frame_roofline_frac"], " / ".join("%.1f" % (v*1e3) for k,v in d["kernel_ms"].items() if k!="-")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
echo "# wisdom.txt"; cat $FFTUP_CACHE_DIR/wisdom.txt
} 2>&1 | tee $OUT/tune.txt
