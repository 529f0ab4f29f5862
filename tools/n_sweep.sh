#!/bin/bash
# SpMM over the dense width N on the two synthetic shapes (one GPU): ms per step and GFLOP/s
for n in 8 16 32 64 128 256; do
  echo "banded nd24k-like N=$n: $(python bench.py --ncols $n --cpu-seconds 0 --no-method2 --steps 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "GFLOP/s")')"
  echo "queen-like 300k rows N=$n: $(timeout -k 10 200 python tools/queen_bench.py 300000 $n | tail -1 | sed 's/.*N=[0-9]*: //' | cut -c1-40)"
done
