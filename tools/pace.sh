#!/bin/bash
# experiment driver: kernel time of the gen-4 SpMM under SBLAS_ABLATE settings given on the command line
for a in "$@"; do
  echo "ablate $a: $(SBLAS_ABLATE=$a python bench.py --cpu-seconds 0 --no-method2 --steps 30 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"], d["ms_per_step"])')"
done
