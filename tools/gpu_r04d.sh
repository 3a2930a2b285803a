#!/bin/bash
TAG=r04_d; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
b() { python bench.py $@ --no-cpu-baseline --no-others --steps 10 --repeats 3 > $OUT/b.json 2>> $OUT/bench.err
  python -c "import json,os; d=json.load(open('$OUT/b.json')); print('%-28s %-44s %9.0f frames/s %.2f us/frame' % (os.environ.get('FFTUP_EXPERIMENT',''), '$*', d['value'], d['ms_per_frame']*1e3), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/bench.txt; }
for e in "u8_plain=0" "u8_plain=1" "u8_plain=0;pairs_per_strip=12" "u8_plain=1;pairs_per_strip=12"; do
FFTUP_EXPERIMENT="row_u8=0;$e" b --preset config3 --fuse-u8-store
done
FFTUP_EXPERIMENT="row_u8=0;u8_plain=1" b --fuse-u8 --fuse-u8-store
FFTUP_EXPERIMENT="row_u8=0" b --fuse-u8 --fuse-u8-store
FFTUP_EXPERIMENT="row_u8=0;u8_plain=1" bash tools/gpu_pmc.sh $TAG/pmc_plain --preset config3 --fuse-u8-store > /dev/null 2>&1
grep -A12 "k_c2r" $OUT/pmc_plain/summary.txt | grep -E "^==|WRITE_SIZE|FETCH_SIZE|SQ_INSTS_VALU|SQ_INSTS_VMEM_WR"
