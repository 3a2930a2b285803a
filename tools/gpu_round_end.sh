#!/bin/bash
# end-of-round evidence: full gpu tests, PMC traffic, bench lines for the BASELINE configs, rocprof kernel stats
TAG=${1:-r01final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_pmc.sh $TAG/pmc > /dev/null 2>&1
python tools/make_traffic_json.py gpurun_out/$TAG/pmc/summary.txt $OUT/hbm_traffic.json > /dev/null && cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
python bench.py > $OUT/bench_fp32.json 2> $OUT/bench.err; cat $OUT/bench_fp32.json
python bench.py --precision 2 --fuse-u8 --no-cpu-baseline > $OUT/bench_fp16_u8.json 2>> $OUT/bench.err; cat $OUT/bench_fp16_u8.json
python bench.py --width 1920 --height 1080 --no-cpu-baseline --frames-per-step 16 --steps 5 > $OUT/bench_1080p.json 2>> $OUT/bench.err; cat $OUT/bench_1080p.json
python bench.py --fuse-u8 --no-cpu-baseline > $OUT/bench_fp32_u8.json 2>> $OUT/bench.err; cat $OUT/bench_fp32_u8.json
python bench.py --precision 1 --no-cpu-baseline --frames-per-step 16 --steps 5 --ring 4 > $OUT/bench_fp64.json 2>> $OUT/bench.err; cat $OUT/bench_fp64.json
python bench.py --host-streamed --no-cpu-baseline --ring 4 > $OUT/bench_host_streamed_fp32.json 2>> $OUT/bench.err; cat $OUT/bench_host_streamed_fp32.json
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --streams 1 > $GRAFT_REPO_ROOT/$OUT/rocprof_bench_streams1.log 2>&1; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/rocprof_bench.log 2>&1)
cat $(find $OUT/prof -name "*kernel_stats.csv" | head -1) | head -8
