#!/bin/bash
# end-of-round evidence on the GPU box:  gpurun --timeout 3000 -- tools/gpu_round_end.sh <tag>   -> gpurun_out/<tag>/
# full GPU suite + smoke, the bench line (with `others`), bench lines of the other configurations, rocprofv3 kernel stats
# (sequential and overlapped), PMC passes per configuration -> hbm_traffic.json.  tools/collect_profiles.sh copies the summaries.
TAG=${1:-rXXfinal}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -s --timeout=900 --durations=20 > $OUT/pytest_gpu.txt 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
python bench.py > $OUT/bench_fp32.json 2> $OUT/bench.err; cat $OUT/bench_fp32.json
python bench.py --preset config3 --no-cpu-baseline > $OUT/bench_fp16_u8.json 2>> $OUT/bench.err
python bench.py --preset config4 --no-cpu-baseline > $OUT/bench_1080p.json 2>> $OUT/bench.err
python bench.py --preset config3 --fuse-u8-store --no-cpu-baseline > $OUT/bench_fp16_u8_u8store.json 2>> $OUT/bench.err
python bench.py --fuse-u8 --fuse-u8-store --no-cpu-baseline > $OUT/bench_fp32_u8_u8store.json 2>> $OUT/bench.err
python bench.py --preset config5 --no-cpu-baseline > $OUT/bench_config5_1gpu.json 2>> $OUT/bench.err
python bench.py --streams 1 --no-cpu-baseline --no-others > $OUT/bench_fp32_streams1.json 2>> $OUT/bench.err
python bench.py --precision 1 --no-cpu-baseline --no-others --no-rccl-check --frames-per-step 64 --steps 5 > $OUT/bench_fp64.json 2>> $OUT/bench.err
python bench.py --host-streamed --no-cpu-baseline --ring 4 > $OUT/bench_host_streamed_fp32.json 2>> $OUT/bench.err
python bench.py --host-streamed --fuse-u8 --fuse-u8-store --no-cpu-baseline --ring 4 > $OUT/bench_host_streamed_u8store.json 2>> $OUT/bench.err
python bench.py --host-streamed --png --no-cpu-baseline --ring 4 > $OUT/bench_host_streamed_png.json 2>> $OUT/bench.err
for f in fp16_u8 1080p fp16_u8_u8store fp32_u8_u8store config5_1gpu fp32_streams1 fp64 host_streamed_fp32 host_streamed_u8store host_streamed_png; do
  python -c "import json,sys; d=json.load(open('$OUT/bench_$f.json')); print('%-24s %9.0f frames/s %.2f us/frame frac %.3f' % ('$f', d['value'], d['ms_per_frame']*1e3, d['frame_roofline_frac']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done
prof() {  # tag, clock_check.py args: rocprofv3 ON fftup_profile_kernels of the DEFAULT plan (ring of 8, three streams: one strip per compute unit), the
          # very launches bench.py's roofline figure comes from (a --streams 1 plan cuts the frame into two strips per unit: another kernel time)
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o bench -- python $R/tools/clock_check.py --mode profile --n 300 ${@:2} > $R/$OUT/rocprof_$1.log 2>&1)
  mkdir -p $R/$OUT/prof_$1; cp $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) $R/$OUT/prof_$1/; rm -rf /tmp/prof_$1      # (the traces stay on the box)
}
prof fp32_s1 --width 2048 --height 1024
prof fp16_u8_s1 --width 2048 --height 1024 --precision 2 --flags 2
prof 1080p_s1 --width 1920 --height 1080
prof fp16_u8_u8store_s1 --width 2048 --height 1024 --precision 2 --flags 34
prof fp32_ordered --width 2048 --height 1024 --ring 1          # the plan fftup_execute's ordered iterations run on (no ring: two strips per unit)
prof fp16_u8_ordered --width 2048 --height 1024 --precision 2 --flags 2 --ring 1
prof 1080p_ordered --width 1920 --height 1080 --ring 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s3 -o bench -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-others --no-live-traffic --no-rccl-check > $R/$OUT/rocprof_fp32_s3.log 2>&1)
mkdir -p $R/$OUT/prof_fp32_s3; cp $(find /tmp/prof_s3 -name "*kernel_stats.csv" | head -1) $R/$OUT/prof_fp32_s3/; rm -rf /tmp/prof_s3
head -5 $(find $OUT/prof_fp32_s1 -name "*kernel_stats.csv" | head -1)
bash tools/gpu_pmc.sh $TAG/pmc_fp32 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/pmc_fp16u8 --preset config3 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/pmc_1080p --preset config4 > /dev/null 2>&1
bash tools/gpu_pmc.sh $TAG/pmc_fp16u8_u8store --preset config3 --fuse-u8-store > /dev/null 2>&1
cp profiles/hbm_traffic.json $OUT/hbm_traffic.json
python tools/make_traffic_json.py $OUT/pmc_fp32/summary.txt $OUT/hbm_traffic.json 2048x1024_p0_planar > /dev/null
python tools/make_traffic_json.py $OUT/pmc_fp16u8/summary.txt $OUT/hbm_traffic.json 2048x1024_p2_u8 > /dev/null
python tools/make_traffic_json.py $OUT/pmc_1080p/summary.txt $OUT/hbm_traffic.json 1920x1080_p0_planar > /dev/null
python tools/make_traffic_json.py $OUT/pmc_fp16u8_u8store/summary.txt $OUT/hbm_traffic.json 2048x1024_p2_u8_u8out > /dev/null
grep -A12 "k_c2r_sharpen_g" $OUT/pmc_fp32/summary.txt | grep -E "==|SQ_INSTS_VALU|FETCH_SIZE|WRITE_SIZE" | head -8
# the bench-line tests against the refreshed figures (in the suite above the committed ones were those of the previous kernels: reported as xfail there)
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
python tools/index_kernel_stats.py 2048x1024_p0_planar $(find $OUT/prof_fp32_s1 -name "*kernel_stats.csv" | head -1) 2048x1024_p2_u8 $(find $OUT/prof_fp16_u8_s1 -name "*kernel_stats.csv" | head -1) \
       1920x1080_p0_planar $(find $OUT/prof_1080p_s1 -name "*kernel_stats.csv" | head -1) 2048x1024_p2_u8_u8out $(find $OUT/prof_fp16_u8_u8store_s1 -name "*kernel_stats.csv" | head -1) > /dev/null
python -m pytest tests/test_gpu_bench.py -q -m gpu -k "bench_line or single_rank_line" -rx > $OUT/pytest_bench_line.txt 2>&1; tail -8 $OUT/pytest_bench_line.txt
