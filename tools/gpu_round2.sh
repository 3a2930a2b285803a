#!/bin/bash
# round-2 evidence on top of tools/gpu_round.sh: size table (plan-time specialised vs generic), rocprof stats of one
# plan-time specialised size:  tools/gpu_round2.sh <tag>
TAG=${1:-r02}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash tools/gpu_round.sh $TAG $2
bash tools/gpu_sizes.sh $TAG/sizes > /dev/null 2>&1; cp $OUT/sizes/sizes.txt $OUT/jit_sizes.txt; cat $OUT/jit_sizes.txt
bash tools/gpu_usizes.sh $TAG/usizes > /dev/null 2>&1; cp $OUT/usizes/usizes.txt $OUT/factors.txt; cat $OUT/factors.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_jit1000_s1 -o bench -- python $R/bench.py --width 1000 --height 1000 --steps 2 --warmup 1 --repeats 1 --frames-per-step 128 --no-cpu-baseline --streams 1 > $R/$OUT/rocprof_jit1000_s1.log 2>&1)
f=$(find $OUT/prof_jit1000_s1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 $f | cut -c1-200
