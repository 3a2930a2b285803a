#!/bin/bash
# copy the judged summaries of a tools/gpu_round_end.sh run into profiles/: tools/collect_profiles.sh <tag> <prefix>   (e.g. r03z r03_z)
T=gpurun_out/$1; P=profiles/$2
for f in $T/bench_*.json; do b=$(basename $f .json); [ -s $f ] && cp $f ${P}_$b.json; done
for d in $T/prof_*; do c=${d##*/prof_}; f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${P}_kernel_stats_$c.csv; done
for d in $T/pmc_*; do c=${d##*/pmc_}; [ -s $d/summary.txt ] && cp $d/summary.txt ${P}_pmc_summary_$c.txt; done
cp $T/pytest_gpu.txt ${P}_pytest_gpu.txt; cp $T/smoke.txt ${P}_smoke.txt
[ -s $T/hbm_traffic.json ] && cp $T/hbm_traffic.json profiles/hbm_traffic.json
ls profiles | grep "^$2" | wc -l
# which kernel-stats summary belongs to which bench configuration (bench.py: roofline.frac_rocprof), stamped with the kernel sources' hash
python tools/index_kernel_stats.py 2048x1024_p0_planar ${P}_kernel_stats_fp32_s1.csv 2048x1024_p2_u8 ${P}_kernel_stats_fp16_u8_s1.csv \
       1920x1080_p0_planar ${P}_kernel_stats_1080p_s1.csv 2048x1024_p2_u8_u8out ${P}_kernel_stats_fp16_u8_u8store_s1.csv > /dev/null
