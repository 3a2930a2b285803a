#!/bin/bash
# copy the judged summaries of a tools/gpu_round.sh run into profiles/: tools/collect_profiles.sh <tag> <prefix>   (e.g. r02f r02_f)
T=gpurun_out/$1; P=profiles/$2
for f in bench_fp32 bench_fp16_u8 bench_1080p bench_fp32_u8 bench_fp64 bench_host_streamed_fp32 bench_fp32_streams1; do
  [ -s $T/$f.json ] && cp $T/$f.json ${P}_$f.json
done
[ -s $T/bench_fp32_g2.json ] && cp $T/bench_fp32_g2.json ${P}_bench_fp32_strips_half_length.json
for c in fp32_s1 fp16_u8_s1 1080p_s1 fp32_s3; do
  f=$(find $T/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${P}_kernel_stats_$c.csv
done
cp $T/pmc_fp32/summary.txt ${P}_pmc_summary_fp32.txt
cp $T/pmc_fp16u8/summary.txt ${P}_pmc_summary_fp16_u8.txt
cp $T/pmc_1080p/summary.txt ${P}_pmc_summary_1080p.txt
cp $T/pytest_gpu.txt ${P}_pytest_gpu.txt; cp $T/smoke.txt ${P}_smoke.txt
cp $T/hbm_traffic.json profiles/hbm_traffic.json
ls profiles | grep "^$2" | wc -l
