#!/bin/bash
# sweep the panel height of the sixth-generation SpMM kernel
for r in "$@"; do
  echo "panel_rows $r: $(SBLAS_SPMM_PANEL_ROWS=$r python bench.py --cpu-seconds 0 --no-extras --steps 30 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["ms_per_step"])')"
done
