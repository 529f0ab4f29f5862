#!/bin/bash
# rocprofv3 kernel stats of the bench under one SBLAS_ABLATE setting ($1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ablate
rm -rf $OUT; mkdir -p $OUT
SBLAS_ABLATE=$1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-method2 > $OUT/bench.json 2> $OUT/err.txt
cat $OUT/run/*/*kernel_stats.csv | cut -c1-60,200-400 | head -8
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/run/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:50], r["Calls"], r["AverageNs"])
PY
