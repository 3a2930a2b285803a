#!/bin/bash
# upscale factors that joined the plan-time compiler last in round 5 (eighths; ratios over 3, 5, 7): specialised vs size-generic kernels,
# frames overlapping on three streams.   gpurun -- tools/gpu_ratio_sizes.sh <tag>   ->  gpurun_out/<tag>/ratio_sizes.txt
TAG=${1:-ratio_sizes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
echo "# bench.py --frames-per-step 256 --steps 3 --repeats 3 --ring 4, fp32; frac = B_alg / t / 8 TB/s; kernels(us): row / column / fused (or C2R / sharpen)"
for cfg in "1920 1080 1.3333334" "960 540 1.3333334" "1600 900 1.6" "1280 720 1.2" "1600 900 1.2" "1000 500 1.4" "1152 648 1.6666666" "768 432 2.6666667" "1280 720 1.125" "2048 1024 1.125" "1024 576 1.875" "1280 720 1.875" "1120 630 1.1428572"; do
  set -- $cfg
  for mode in jit generic; do
    if [ $mode = generic ]; then FL="--generic"; else FL=""; fi
    python bench.py --width $1 --height $2 --upscale $3 --no-cpu-baseline --no-others --no-rccl-check --no-live-traffic --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 $FL > $OUT/b_$1x$2_u$3_$mode.json 2>> $OUT/err.txt
    python - $OUT/b_$1x$2_u$3_$mode.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-44s %-9s %8.1f us/frame  %8.0f frames/s  frac %.3f  kernels(us) %s  | %s" % (d["config"]["workload"].split(",")[0], d["config"]["kernels"], d["ms_per_frame"]*1e3, d["value"], d["frame_roofline_frac"], " / ".join("%.1f" % (v*1e3) for k,v in d["kernel_ms"].items() if k!="-"), d["config"]["plan"][25:70]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
done
} | tee $OUT/ratio_sizes.txt
