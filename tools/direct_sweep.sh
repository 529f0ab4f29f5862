#!/bin/bash
# direct DPP kernel on the banded bench matrix under the experimental occupancy / mapping switches
for n in 64 128; do for m in contiguous interleave; do for l in 0 90000; do
  echo "N=$n map=$m ldspad=$l: $(SBLAS_SPMM_VARIANT=dpp SBLAS_DIRECT_MAP=$m SBLAS_DIRECT_LDS=$l python bench.py --ncols $n --cpu-seconds 0 --no-extras --steps 20 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done; done; done
