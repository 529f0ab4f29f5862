#!/bin/bash
for a in 0 2 8 10; do
  SBLAS_ABLATE=$a SBLAS_SPMM_VARIANT=win3 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('ablate=$a (2=no consumer math, 8=no tile loads) kernel=%.4f ms' % (r['kernel_ms']))"
done
