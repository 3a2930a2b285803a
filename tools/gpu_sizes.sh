#!/bin/bash
# frame time / algorithmic-byte fraction over a list of sizes, run-time specialised plans vs the size-generic kernels
# (FFTUP_JIT=0):  tools/gpu_sizes.sh <tag>   ->  gpurun_out/<tag>/sizes.txt
TAG=${1:-sizes}; OUT=gpurun_out/$TAG; mkdir -p $OUT
line() {  # json label
python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-46s %-9s %8.1f us/frame  %8.0f frames/s  frac %.3f  kernels(us) %s" % (d["config"]["workload"].split(",")[0], sys.argv[2], d["ms_per_frame"]*1e3, d["value"], d["frame_roofline_frac"], " / ".join("%.1f" % (v*1e3) for k,v in d["kernel_ms"].items() if k!="-")))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
{
echo "# bench.py --frames-per-step 256 --steps 3 --repeats 3 --ring 4, fp32 planar in / fp32 planes out; frac = B_alg / t / 8 TB/s"
echo "# 'plan-time' = kernels instantiated for the size through hipRTC at plan creation (csrc/jit.hpp), 'tuned' = ahead-of-time kernels, 'generic' = FFTUP_FLAG_GENERIC_KERNELS"
for cfg in "640 480" "720 480" "720 576" "800 600" "1000 1000" "1024 768" "1280 720" "1280 1024" "1440 900" "1600 900" "1920 1080" "1920 1200" "2000 2000" "2048 1024" "2560 1440" "3584 2016" "3840 2160" "4096 2048"; do
  set -- $cfg
  for mode in jit generic; do
    if [ $mode = generic ]; then FL="--generic"; else FL=""; fi
    python bench.py --width $1 --height $2 --no-cpu-baseline --no-others --no-rccl-check --no-live-traffic --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 $FL > $OUT/b_$1x$2_$mode.json 2>> $OUT/err.txt
    k=$(python -c "import json;d=json.loads(open('$OUT/b_$1x$2_$mode.json').read().strip().splitlines()[-1]);print(d['config']['kernels'])" 2>/dev/null)
    line $OUT/b_$1x$2_$mode.json "$k"
  done
done
echo "# other upscale factors"
for cfg in "1920 1080 1.5" "2048 1024 1.5" "2048 1024 3" "960 540 4"; do
  set -- $cfg
  python bench.py --width $1 --height $2 --upscale $3 --no-cpu-baseline --no-others --no-rccl-check --no-live-traffic --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 > $OUT/b_$1x$2_u$3.json 2>> $OUT/err.txt
  k=$(python -c "import json;d=json.loads(open('$OUT/b_$1x$2_u$3.json').read().strip().splitlines()[-1]);print(d['config']['kernels'])" 2>/dev/null)
  line $OUT/b_$1x$2_u$3.json "$k"
done
} | tee $OUT/sizes.txt
