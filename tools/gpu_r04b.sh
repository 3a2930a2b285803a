#!/bin/bash
# round 4, step b: new u8 row kernel + transposed / XCD-colocated u8 store: parity, kernel times, PMC write sizes
TAG=r04_b; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=900 -k "u8 or full_size_vs_oracle or recorded or strip_length or crops or config1" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
for a in "--preset config3" "--preset config3 --fuse-u8-store" "--fuse-u8 --fuse-u8-store" ""; do
  python bench.py $a --no-cpu-baseline --no-others --steps 10 --repeats 3 > $OUT/b.json 2>> $OUT/bench.err
  python -c "import json; d=json.load(open('$OUT/b.json')); print('%-44s %9.0f frames/s %.2f us/frame' % ('$a', d['value'], d['ms_per_frame']*1e3), {k: round(v*1e3,2) for k,v in d['kernel_ms_isolated'].items()} if 'kernel_ms_isolated' in d else '', {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/bench.txt
done
bash tools/gpu_pmc.sh $TAG/pmc_fp16u8_u8store --preset config3 --fuse-u8-store > /dev/null 2>&1
grep -E "^==|WRITE_SIZE|FETCH_SIZE|SQ_INSTS_VALU|SQ_INSTS_VMEM_WR" $OUT/pmc_fp16u8_u8store/summary.txt
prof() { (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-others ${@:2} > $R/$OUT/rocprof_$1.log 2>&1)
  mkdir -p $R/$OUT/prof_$1; cp $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) $R/$OUT/prof_$1/; rm -rf /tmp/prof_$1; head -5 $R/$OUT/prof_$1/*kernel_stats.csv | cut -c1-150; }
prof fp16_u8_s1 --preset config3 --streams 1
prof fp16_u8_u8store_s1 --preset config3 --fuse-u8-store --streams 1
