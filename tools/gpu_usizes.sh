#!/bin/bash
# integer and half-integer upscale factors: plan-time specialised vs size-generic kernels.  tools/gpu_usizes.sh <tag>
TAG=${1:-usizes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
echo "# bench.py --frames-per-step 256 --steps 3 --repeats 3 --ring 4, fp32; frac = B_alg / t / 8 TB/s; kernels(us): row / column / fused (or C2R / sharpen)"
for cfg in "1280 720 1.5" "1920 1080 1.5" "2560 1440 1.5" "2048 1024 1.5" "640 480 1.5" "1280 720 2.5" "1920 1080 2.5" "640 480 3" "640 480 4" "960 540 4" "1280 720 3" "1920 1080 3" "1920 1080 4" "2048 1024 3" "2048 1024 4" "1024 512 4" "1024 512 8" "1600 900 5"; do
  set -- $cfg
  for mode in jit generic; do
    if [ $mode = generic ]; then FL="--generic"; else FL=""; fi
    python bench.py --width $1 --height $2 --upscale $3 --no-cpu-baseline --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 $FL > $OUT/b_$1x$2_u$3_$mode.json 2>> $OUT/err.txt
    python - $OUT/b_$1x$2_u$3_$mode.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-46s %-9s %8.1f us/frame  %8.0f frames/s  frac %.3f  kernels(us) %s" % (d["config"]["workload"].split(",")[0], d["config"]["kernels"], d["ms_per_frame"]*1e3, d["value"], d["frame_roofline_frac"], " / ".join("%.1f" % (v*1e3) for k,v in d["kernel_ms"].items() if k!="-")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
} | tee $OUT/usizes.txt
