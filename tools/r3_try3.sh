set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "every_column_count or windowed or config4 or census or mfma or row_block" > gpurun_out/r3_w6nh_tests.txt 2>&1 || (tail -30 gpurun_out/r3_w6nh_tests.txt; exit 1)
tail -2 gpurun_out/r3_w6nh_tests.txt
run() { n=$1; tag=$2; python bench.py --ncols $n --no-extras --cpu-seconds 0 --steps 50 > gpurun_out/r3_n${n}_$tag.json 2> gpurun_out/r3_n${n}_$tag.err || (tail -5 gpurun_out/r3_n${n}_$tag.err; exit 1); python - <<PY
import json;d=json.loads(open('gpurun_out/r3_n${n}_$tag.json').read().strip().splitlines()[-1]);print($n, '$tag', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['panels'], d['oracle_check'])
PY
}
run 128 nh2; SBLAS_TUNE=0,1,0,0 run 128 nh1
run 256 nh2; SBLAS_TUNE=0,1,0,0 run 256 nh1
run 64 base
