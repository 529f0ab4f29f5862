#!/bin/bash
# SpMM over the dense width N on the two synthetic shapes (one GPU): ms per step and GFLOP/s
for n in 8 16 32 64 128 256; do
  echo "banded nd24k-like N=$n: $(python bench.py --ncols $n --cpu-seconds 0 --no-extras --steps 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "GFLOP/s")')"
  echo "queen-like 300k rows N=$n: $(timeout -k 10 200 python tools/spmm_shapes.py queen:300000 --n $n --steps 5 --rounds 1 | tail -1 | cut -c1-120)"
done
