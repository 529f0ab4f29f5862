set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_drivers.py tests/test_gpu_plan.py -m gpu -x -q > gpurun_out/r3_plan_tests.txt 2>&1 || (tail -30 gpurun_out/r3_plan_tests.txt; exit 1)
tail -2 gpurun_out/r3_plan_tests.txt
python bench.py --cpu-seconds 0 > gpurun_out/r3_bench_plan.json 2> gpurun_out/r3_bench_plan.err || (tail -20 gpurun_out/r3_bench_plan.err; exit 1)
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench_plan.json').read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], 'planned', d['planned'])
print('widths', {k:(v['ms_per_step']) for k,v in d['method1_widths'].items() if k!='note'})
for k,v in d['secondary'].items():
    print(k, v['ms_per_step'], v['planned_ms_per_step'], v['plan'], v.get('method2_rank_share'))
for k,v in d.get('product_merge',{}).items(): print(k, v.get('ms_per_step'), v.get('oracle_check'))
PY
