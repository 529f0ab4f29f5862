set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_comm_stub.py -x -q > gpurun_out/r3_stub2.txt 2>&1 || (tail -30 gpurun_out/r3_stub2.txt; exit 1)
tail -2 gpurun_out/r3_stub2.txt
python bench.py --product-merge-child 0 --merge-steps 4 > gpurun_out/r3_pm.txt 2> gpurun_out/r3_pm.err || (tail -20 gpurun_out/r3_pm.err; exit 1)
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_pm.txt').read().splitlines() if x.startswith('PRODUCT_MERGE_JSON ')][-1]
d=json.loads(l[len('PRODUCT_MERGE_JSON '):])
for k,v in d.items(): print(k, v['ms_per_step'], v['ms_spmm_max_over_ranks'], v['ms_merge_max_over_ranks'], v['oracle_check'], v.get('timeline_rank0_ms'))
PY
