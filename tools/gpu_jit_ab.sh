#!/bin/bash
# A/B of pinned factorizations of the run-time specialised plans: tools/gpu_jit_ab.sh <tag>
TAG=${1:-jitab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() {   # W H label env...
  local W=$1 H=$2 L=$3; shift 3
  env "$@" python bench.py --width $W --height $H --no-cpu-baseline --steps 3 --warmup 1 --repeats 3 --frames-per-step 256 --ring 4 > $OUT/b_${W}x${H}_$L.json 2>> $OUT/err.txt
  python - $OUT/b_${W}x${H}_$L.json "$L $*" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-26s %-44s %7.1f us/frame frac %.3f %s %s" % (d["config"]["workload"][:26], sys.argv[2][:44], d["ms_per_frame"]*1e3, d["frame_roofline_frac"], d["config"]["kernels"], {k: round(v*1e3,1) for k,v in d["kernel_ms"].items() if k!="-"}))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
run 2000 2000 default X=1
run 2000 2000 col8 FFTUP_JIT_COL=8,5,5,10
run 2000 2000 col10 FFTUP_JIT_COL=10,5,4,10
run 2000 2000 row10 FFTUP_JIT_ROW=10,2,10,10
run 3584 2016 default X=1
run 3584 2016 row8 FFTUP_JIT_ROW=8,8,8,7
run 3584 2016 col FFTUP_JIT_COL=9,4,7,8
run 3200 1800 default X=1
run 3200 1800 row8 FFTUP_JIT_ROW=8,5,8,10
run 3840 2160 default X=1
run 1600 900 default X=1
run 2560 1440 default X=1
run 1440 900 default X=1
run 800 600 default X=1
run 720 480 default X=1
run 640 480 default X=1
