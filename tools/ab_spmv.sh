# A/B of library builds on one box, SpMV bench: tools/ab_spmv.sh lib1 lib2 ... (files under gpurun_ab/), two interleaved rounds
cd $GRAFT_REPO_ROOT
cp s-blas_amd/lib/libsblas_hip.so /tmp/orig.so
for round in 1 2; do for l in "$@"; do
  cp gpurun_ab/$l s-blas_amd/lib/libsblas_hip.so
  python bench.py --op spmv --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('oracle_check'))"
done; done
cp /tmp/orig.so s-blas_amd/lib/libsblas_hip.so
