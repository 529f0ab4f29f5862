#!/bin/bash
for a in 0 1 2 3; do
  SBLAS_ABLATE=$a SBLAS_SPMM_VARIANT=win2 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('ablate=$a kernel=%.4f ms check=%s' % (r['kernel_ms'], d['oracle_check']))"
done
