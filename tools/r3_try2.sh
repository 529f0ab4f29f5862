set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_narrow.py -x -q > gpurun_out/r3_narrow_tests.txt 2>&1 || (tail -30 gpurun_out/r3_narrow_tests.txt; exit 1)
tail -2 gpurun_out/r3_narrow_tests.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --fold-ranks --no-extras --no-method2 > gpurun_out/r3_fold2.json 2> gpurun_out/r3_fold2.err || (tail -20 gpurun_out/r3_fold2.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_fold2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('method1_strong'))
PY
