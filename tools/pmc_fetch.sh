#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate runs) on shapes of tools/spmm_shapes.py:  tools/pmc_fetch.sh OUTNAME "ARGS" ["ARGS" ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT; mkdir -p $OUT
shift
i=0
for args in "$@"; do
  i=$((i+1))
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/run${i}_set_$c -- python3 $R/tools/spmm_shapes.py $args --rounds 1 --steps 2 > $OUT/run${i}_$c.log 2>&1
    echo "run$i $c rc=$? ($args)"
  done
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -v "classify\|stage_\|dense_to" $OUT/summary.txt
