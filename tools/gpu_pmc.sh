#!/bin/bash
# PMC counter passes (each in its own rocprofv3 run, kernel-trace only) for the bench workload.
# usage: tools/gpu_pmc.sh [tag] [extra bench args]
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --repeats 1 --frames-per-step 8 --profile-iters 2 --no-cpu-baseline --no-others --no-live-traffic --no-rccl-check $@"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "FETCH_SIZE TCC_HIT_sum" \
  "WRITE_SIZE TCC_MISS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- $BENCH > $OUT/pass$i.log 2>&1
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/pass[0-9]* $OUT/counters_list.txt      # (raw counter files: tens of MB; gpurun_out/ comes back only below 64 MiB)
cat $OUT/summary.txt
