#!/bin/bash
# kernel time of the SpMV variants on the bench matrix
for v in "$@"; do
  echo "spmv $v: $(SBLAS_SPMV_VARIANT=$v python bench.py --op spmv --steps 100 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["roofline"]["frac"])')"
done
