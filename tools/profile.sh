#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command (summaries are copied into profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmm -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-extras > $OUT/spmm_bench.json 2> $OUT/spmm.err
echo "spmm rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmv -- python3 $R/bench.py --op spmv --steps 50 --warmup 5 > $OUT/spmv_bench.json 2> $OUT/spmv.err
echo "spmv rc=$?"
# HBM traffic counters, one pass each (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras > /dev/null 2> $OUT/pmc_$c.err
  echo "$c rc=$?"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcspmv_$c -- python3 $R/bench.py --op spmv --steps 5 --warmup 1 > /dev/null 2> $OUT/pmcspmv_$c.err
  echo "spmv $c rc=$?"
done
# the traffic figures the bench line quotes must come from THESE sources: summarise the counter passes into profiles/ of this
# copy first (the caller runs collect_profiles.py again on the merged gpurun_out/ to commit them)
cd $R && python tools/collect_profiles.py r02 > $OUT/collect.log 2>&1
cd $R && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; cat $OUT/bench_default.json
python bench.py --op spmv --steps 100 > $OUT/bench_spmv.json 2>/dev/null; cat $OUT/bench_spmv.json
