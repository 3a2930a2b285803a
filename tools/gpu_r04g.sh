#!/bin/bash
TAG=r04_g; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
FFTUP_EXPERIMENT=colp=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout=900 -k "full_size_vs_oracle and 2048" > $OUT/pytest.txt 2>&1; tail -2 $OUT/pytest.txt
b() { python bench.py $@ --no-cpu-baseline --no-others --steps 10 --repeats 3 > $OUT/b.json 2>> $OUT/bench.err
  python -c "import json,os; d=json.load(open('$OUT/b.json')); print('%-10s %-30s %9.0f frames/s %.2f us/frame %s W' % (os.environ.get('FFTUP_EXPERIMENT',''), '$*', d['value'], d['ms_per_frame']*1e3, d['power']['socket_power_w_median']), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/bench.txt; }
for rep in 1 2; do for v in 0 1; do FFTUP_EXPERIMENT=colp=$v b; FFTUP_EXPERIMENT=colp=$v b --streams 1; FFTUP_EXPERIMENT=colp=$v b --preset config3; done; done
prof() { (cd /tmp && FFTUP_EXPERIMENT=$3 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o bench -- python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-others --streams 1 $2 > $R/$OUT/rocprof_$1.log 2>&1)
  mkdir -p $R/$OUT/prof_$1; cp $(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1) $R/$OUT/prof_$1/; rm -rf /tmp/prof_$1; head -4 $R/$OUT/prof_$1/*kernel_stats.csv | cut -c1-150; }
prof fp32_colp1 "" colp=1
prof fp32_colp0 "" colp=0
bash tools/gpu_clock_check.sh $TAG
