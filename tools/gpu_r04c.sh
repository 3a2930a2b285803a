#!/bin/bash
TAG=r04_c; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s --timeout=900 -k "u8" > $OUT/pytest.txt 2>&1; grep -E "U8STORE|passed|failed|FAILED" $OUT/pytest.txt | tail -15
b() { python bench.py $@ --no-cpu-baseline --no-others --steps 10 --repeats 3 > $OUT/b.json 2>> $OUT/bench.err
  python -c "import json,os; d=json.load(open('$OUT/b.json')); print('%-10s %-44s %9.0f frames/s %.2f us/frame' % (os.environ.get('FFTUP_EXPERIMENT',''), '$*', d['value'], d['ms_per_frame']*1e3), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/bench.txt; }
for v in 0 1 2; do FFTUP_EXPERIMENT=row_u8=$v b --preset config3; done
for v in 0 2; do FFTUP_EXPERIMENT=row_u8=$v b --fuse-u8; done
b --preset config3 --fuse-u8-store
b --fuse-u8 --fuse-u8-store
b --preset config4 --fuse-u8 --fuse-u8-store
bash tools/gpu_pmc.sh $TAG/pmc_fp16u8_u8store --preset config3 --fuse-u8-store > /dev/null 2>&1
grep -E "^==|WRITE_SIZE|FETCH_SIZE|SQ_INSTS_VALU|SQ_INSTS_VMEM_WR" $OUT/pmc_fp16u8_u8store/summary.txt | grep -v copyBuffer
prof() { (cd /tmp && FFTUP_EXPERIMENT=$3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-others --streams 1 $2 > $R/$OUT/rocprof_$1.log 2>&1)
  mkdir -p $R/$OUT/prof_$1; cp $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) $R/$OUT/prof_$1/; rm -rf /tmp/prof_$1; head -4 $R/$OUT/prof_$1/*kernel_stats.csv | cut -c1-150; }
prof u8store "--preset config3 --fuse-u8-store" row_u8=2
prof rowu8_0 "--preset config3" row_u8=0
prof rowu8_1 "--preset config3" row_u8=1
