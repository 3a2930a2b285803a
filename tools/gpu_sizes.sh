#!/bin/bash
# frame time / algorithmic-byte fraction over a list of sizes (generic and tuned plans): tools/gpu_sizes.sh <tag>
TAG=${1:-sizes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for cfg in "640 480 2" "720 480 2" "1000 1000 2" "1280 720 2" "1920 1080 2" "1920 1080 1.5" "2048 1024 1.5" "2048 1024 3" "3840 2160 2" "4096 2048 2" "960 540 4" "1024 1024 2" "2560 1440 1.5"; do
  set -- $cfg
  python bench.py --width $1 --height $2 --upscale $3 --no-cpu-baseline --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 ${EXTRA} > $OUT/b_$1x$2_u$3.json 2>> $OUT/err.txt
  python - $OUT/b_$1x$2_u$3.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s %8.1f us/frame  frac %.3f  kernels %s  %s" % (d["config"]["workload"][:44], d["ms_per_frame"]*1e3, d["frame_roofline_frac"], d["config"]["kernels"], {k: round(v*1e3,1) for k,v in d["kernel_ms"].items() if k!="-"}))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
