#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_tcc
rm -rf $OUT; mkdir -p $OUT
for v in ${VARIANTS:-auto}; do
 i=0
 for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "WRITE_SIZE" "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  i=$((i+1))
  SBLAS_SPMM_VARIANT=$v rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${v}_set$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-extras > $OUT/${v}_set$i.log 2>&1
  echo "$v set$i rc=$?"
 done
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
