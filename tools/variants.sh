#!/bin/bash
# A/B the stage-2 SpMM variants on the bench workload (one process each; kernel_ms from HIP events)
for v in ${VARIANTS:-win4 win3 dpp}; do
  SBLAS_SPMM_VARIANT=$v python bench.py --steps ${STEPS:-20} --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('%-10s value=%9.1f GF/s  step=%.4f ms  kernel=%.4f ms  frac=%.4f' % ('$v', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))"
done
