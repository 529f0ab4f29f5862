#!/bin/bash
# bits: 2 no consumer math, 8 no tile loads, 16 no tile-loop barriers, 32 no A loads (synthetic columns), 64 no window shift
for a in 0 10 26 42 58 74 122; do
  SBLAS_ABLATE=$a SBLAS_SPMM_VARIANT=win3 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('ablate=%3d kernel=%.4f ms' % ($a, r['kernel_ms']))"
done
