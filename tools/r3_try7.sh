set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_narrow.py -x -q > gpurun_out/r3_lpe.txt 2>&1 || (tail -30 gpurun_out/r3_lpe.txt; exit 1)
tail -2 gpurun_out/r3_lpe.txt
run() { n=$1; tag=$2; python bench.py --ncols $n --no-extras --cpu-seconds 0 --steps 100 > gpurun_out/r3_n${n}_$tag.json 2> gpurun_out/r3_n${n}_$tag.err || (tail -5 gpurun_out/r3_n${n}_$tag.err; exit 1); python - <<PY
import json;d=json.loads(open('gpurun_out/r3_n${n}_$tag.json').read().strip().splitlines()[-1]);print($n, '$tag', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['panels'], d['oracle_check'])
PY
}
run 16 auto
SBLAS_SPMM_PANEL_ROWS=96,2 run 16 g2
SBLAS_SPMM_PANEL_ROWS=144,3 run 16 g3
run 16 auto
