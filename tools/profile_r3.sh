#!/bin/bash
# Round-3 profiles: rocprofv3 --kernel-trace --stats of the default bench command and of the narrow / wide widths,
# FETCH_SIZE / WRITE_SIZE passes (separate runs) per width, kernel-stats of the SpMV medium- / short-row shapes.
# Everything lands in gpurun_out/prof; tools/collect_profiles.py r03 copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmm -- python3 $R/bench.py --steps 50 --warmup 5 --cpu-seconds 0 --no-extras > $OUT/spmm_bench.json 2> $OUT/spmm.err
echo "spmm rc=$?"
for n in 8 16 32 128 256; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmm_n$n -- python3 $R/bench.py --ncols $n --steps 50 --warmup 5 --cpu-seconds 0 --no-extras > $OUT/spmm_n${n}_bench.json 2> $OUT/spmm_n$n.err
  echo "spmm n=$n rc=$?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmv -- python3 $R/bench.py --op spmv --steps 50 --warmup 5 > $OUT/spmv_bench.json 2> $OUT/spmv.err
echo "spmv rc=$?"
# SpMV medium / short rows (the figures DESIGN quotes): Queen-like rows (segmented kernel), stencil-like rows (stream kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmv_queen -- python3 $R/tools/queen_spmv.py 1000000 > $OUT/spmv_queen.txt 2> $OUT/spmv_queen.err
echo "spmv queen rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/spmv_short -- python3 $R/tools/spmv_shapes.py banded:1000000:13:40 banded:1000000:27:60 banded:600000:48:2000 --rounds 1 --steps 20 > $OUT/spmv_short.txt 2> $OUT/spmv_short.err
echo "spmv short rc=$?"
# HBM traffic counters, one pass each (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-settle > /dev/null 2> $OUT/pmc_$c.err
  echo "$c rc=$?"
  for n in 8 16 32 128; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcn${n}_$c -- python3 $R/bench.py --ncols $n --steps 5 --warmup 1 --cpu-seconds 0 --no-extras --no-settle > /dev/null 2> $OUT/pmcn${n}_$c.err
    echo "n=$n $c rc=$?"
  done
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcspmv_$c -- python3 $R/bench.py --op spmv --steps 5 --warmup 1 > /dev/null 2> $OUT/pmcspmv_$c.err
  echo "spmv $c rc=$?"
done
cd $R && python tools/collect_profiles.py r03 > $OUT/collect.log 2>&1
cd $R && python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench_default.json
python bench.py --op spmv --steps 100 > $OUT/bench_spmv.json 2>/dev/null; cat $OUT/bench_spmv.json
