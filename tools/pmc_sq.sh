#!/bin/bash
# SQ / GRBM counter passes for the gen-4 SpMM kernel under the SBLAS_ABLATE settings given as arguments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq
mkdir -p $OUT
for a in default; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES" \
             "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/a${a}_set$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > $OUT/a${a}_set$i.log 2>&1
    echo "ablate $a set$i rc=$?"
  done
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
