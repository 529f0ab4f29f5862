#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
for v in ${VARIANTS:-win3 dpp}; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_WAVE32_LDS" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    SBLAS_SPMM_VARIANT=$v rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${v}_set$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/${v}_set$i.log 2>&1
    echo "$v set$i rc=$?"
  done
done
