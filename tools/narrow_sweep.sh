#!/bin/bash
# N <= 32: narrow kernels (ldbt = 8/16/32) against the 64-column kernels on zero-padded columns
for n in 8 16 32; do for m in 0 64; do
  echo "banded N=$n min_ldbt=$m: $(SBLAS_SPMM_MIN_LDBT=$m python bench.py --ncols $n --cpu-seconds 0 --no-extras --steps 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  echo "queen  N=$n min_ldbt=$m: $(SBLAS_SPMM_MIN_LDBT=$m timeout -k 10 200 python tools/spmm_shapes.py queen:300000 --n $n --steps 5 --rounds 1 | tail -1 | cut -c1-120)"
done; done
