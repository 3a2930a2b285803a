export FFTUP_CACHE_DIR=/tmp/aotab; rm -rf $FFTUP_CACHE_DIR
for cfg in "2048 1024" "1920 1080" "1280 720" "1024 512"; do
  set -- $cfg
  for mode in aot jit jit_tuned; do
    unset FFTUP_AOT FFTUP_JIT_TUNE FFTUP_JIT_VERBOSE
    [ $mode != aot ] && export FFTUP_AOT=0
    [ $mode = jit_tuned ] && export FFTUP_JIT_TUNE=1 FFTUP_JIT_VERBOSE=1
    python bench.py --width $1 --height $2 --no-cpu-baseline --steps 5 --warmup 1 --repeats 3 --frames-per-step 512 2> /tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-30s %-10s %7.1f us/frame frac %.3f %s %s' % (d['config']['workload'][:30], '$mode', d['ms_per_frame']*1e3, d['frame_roofline_frac'], d['config']['kernels'], ' / '.join('%.1f'%(v*1e3) for k,v in d['kernel_ms'].items() if k!='-')))"
    grep "fftup: tuning" /tmp/err.txt | sed 's/^/    /'
  done
done
cat $FFTUP_CACHE_DIR/wisdom.txt
