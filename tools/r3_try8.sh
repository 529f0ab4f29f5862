set -e
cd $GRAFT_REPO_ROOT
run() { n=$1; tag=$2; python bench.py --ncols $n --no-extras --cpu-seconds 0 --steps 100 > gpurun_out/r3_sw_$tag.json 2> gpurun_out/r3_sw_$tag.err || (tail -5 gpurun_out/r3_sw_$tag.err; exit 1); python - <<PY
import json;d=json.loads(open('gpurun_out/r3_sw_$tag.json').read().strip().splitlines()[-1]);print($n, '$tag', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['panels'], d['oracle_check'])
PY
}
run 64 auto
for pr in 144,3 132,3 120,3 88,2 80,2 72,2 96,2; do SBLAS_SPMM_PANEL_ROWS=$pr run 64 p$pr; done
run 64 auto
