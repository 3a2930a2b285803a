#!/bin/bash
# evidence for profiles/: PMC traffic per config, bench lines for the BASELINE configs, rocprofv3 kernel stats, gpu tests
# usage: tools/gpu_round.sh <tag> [quick]
TAG=${1:-r02}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# ---- PMC passes per configuration -> keyed traffic json
bash tools/gpu_pmc.sh $TAG/pmc_fp32 > /dev/null 2>&1
python tools/make_traffic_json.py $OUT/pmc_fp32/summary.txt profiles/hbm_traffic.json 2048x1024_p0_planar > /dev/null
bash tools/gpu_pmc.sh $TAG/pmc_fp16u8 --precision 2 --fuse-u8 > /dev/null 2>&1
python tools/make_traffic_json.py $OUT/pmc_fp16u8/summary.txt profiles/hbm_traffic.json 2048x1024_p2_u8 > /dev/null
bash tools/gpu_pmc.sh $TAG/pmc_1080p --width 1920 --height 1080 > /dev/null 2>&1
python tools/make_traffic_json.py $OUT/pmc_1080p/summary.txt profiles/hbm_traffic.json 1920x1080_p0_planar > /dev/null
cp profiles/hbm_traffic.json $OUT/hbm_traffic.json
# ---- bench lines
python bench.py > $OUT/bench_fp32.json 2> $OUT/bench.err; cat $OUT/bench_fp32.json
python bench.py --preset config3 --no-cpu-baseline > $OUT/bench_fp16_u8.json 2>> $OUT/bench.err; cat $OUT/bench_fp16_u8.json
python bench.py --preset config4 --no-cpu-baseline > $OUT/bench_1080p.json 2>> $OUT/bench.err; cat $OUT/bench_1080p.json
if [ "$2" != "quick" ]; then
python bench.py --fuse-u8 --no-cpu-baseline --steps 5 > $OUT/bench_fp32_u8.json 2>> $OUT/bench.err
python bench.py --precision 1 --no-cpu-baseline --frames-per-step 64 --steps 5 --ring 4 > $OUT/bench_fp64.json 2>> $OUT/bench.err
python bench.py --host-streamed --no-cpu-baseline --ring 4 --frames-per-step 64 --steps 5 > $OUT/bench_host_streamed_fp32.json 2>> $OUT/bench.err
python bench.py --streams 1 --no-cpu-baseline --steps 5 > $OUT/bench_fp32_streams1.json 2>> $OUT/bench.err
FFTUP_G_PER_CU=2 python bench.py --no-cpu-baseline --steps 5 > $OUT/bench_fp32_g2.json 2>> $OUT/bench.err
fi
# ---- rocprofv3 kernel stats: sequential launches (streams 1) and the default
for cfg in "fp32:" "fp16_u8:--preset config3" "1080p:--preset config4"; do
  name=${cfg%%:*}; a=${cfg#*:}
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_${name}_s1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --frames-per-step 128 --no-cpu-baseline --streams 1 $a > $R/$OUT/rocprof_${name}_s1.log 2>&1)
done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_fp32_s3 -o bench -- python $R/bench.py --steps 2 --warmup 1 --repeats 1 --frames-per-step 128 --no-cpu-baseline > $R/$OUT/rocprof_fp32_s3.log 2>&1)
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -5 $f | cut -c1-160; done
if [ "$2" != "quick" ]; then
timeout 1500 python -m pytest tests -m gpu -q -s --timeout=900 > $OUT/pytest_gpu.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
fi
