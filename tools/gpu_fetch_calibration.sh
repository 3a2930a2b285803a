#!/bin/bash
# FETCH_SIZE calibration of the frame's kernels on THEMSELVES:  gpurun -- tools/gpu_fetch_calibration.sh <tag>
# fftup_profile_kernels of the headline plan with 1 GB written in front of every kernel launch (test build of the library,
# FFTUP_EXPERIMENT evict_mb=1024): each kernel then finds nothing of its predecessor's output in the L2s or the Infinity Cache and
# reads a KNOWN number of bytes -- row pass: the planar input, 3 x 2048 x 1024 x 4 = 25.17 MB; column pass: S1 = 25.26 MB;
# fused kernel: S1 + S2 rows, one halo pair per strip (13/12) = 54.7 MB.  The counters per kernel, beside the same run without the fill.
TAG=${1:-fetch_cal}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in evict plain; do
  if [ $mode = evict ]; then export FFTUP_LIBRARY=$GRAFT_REPO_ROOT/vkresample_amd/libfftup_knobs.so FFTUP_EXPERIMENT=evict_mb=1024; else unset FFTUP_LIBRARY FFTUP_EXPERIMENT; fi
  i=0; mkdir -p $OUT/$mode
  for SET in "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_READ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$mode/pass$i -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/clock_check.py --mode profile --n 20 --width 2048 --height 1024 "${@:2}" > $OUT/$mode/pass$i.log 2>&1
  done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$mode > $OUT/summary_$mode.txt 2>&1
  rm -rf $OUT/$mode
done
grep -A12 -E "k_row_r2c|k_col_v|k_c2r_sharpen" $OUT/summary_evict.txt | grep -v fill
echo ---- without the fill; grep -A12 -E "k_row_r2c|k_col_v|k_c2r_sharpen" $OUT/summary_plain.txt
