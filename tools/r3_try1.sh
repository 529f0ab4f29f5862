set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "every_column_count or narrow or leading or nonfinite or method1 or rows_not_multiple" 2>&1 | tail -3
run() { n=$1; tag=$2; python bench.py --ncols $n --no-extras --cpu-seconds 0 --steps 50 > gpurun_out/r3_n${n}_$tag.json 2> gpurun_out/r3_n${n}_$tag.err || (tail -5 gpurun_out/r3_n${n}_$tag.err; exit 1); python -c "
import json;d=json.load(open('gpurun_out/r3_n${n}_$tag.json'));print($n, '$tag', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['panels'], d['oracle_check'])"; }
for cp in 1 2 4; do SBLAS_TUNE=$cp,0,0,0 run 8 cp$cp; done
for cp in 1 2; do SBLAS_TUNE=$cp,0,0,0 run 16 cp$cp; done
SBLAS_TUNE=1,0,0,0 SBLAS_SPMM_PANEL_ROWS=96,2 run 8 cp1g2
