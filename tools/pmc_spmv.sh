#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_spmv
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE"; do
  i=$((i+1))
  SBLAS_SPMV_VARIANT=plain rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/plain_set$i -- python3 $R/bench.py --op spmv --steps 5 --warmup 1 > $OUT/set$i.log 2>&1
  echo "set$i rc=$?"
done
