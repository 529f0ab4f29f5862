#!/bin/bash
# PMC passes (counters only; no tracing flags besides kernel-trace) on a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
for v in ${VARIANTS:-direct win32}; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    SBLAS_SPMM_VARIANT=$v rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${v}_set$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/${v}_set$i.log 2>&1
    echo "$v set$i rc=$?"
  done
done
