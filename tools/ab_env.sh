# tools/ab_env.sh <ncols> "<lib> <ENV=val ...>" ... : A/B of (library build, environment) pairs on one box, two interleaved rounds
cd $GRAFT_REPO_ROOT
n=$1; shift
cp s-blas_amd/lib/libsblas_hip.so /tmp/orig.so
specs=("$@"); for round in 1 2; do for spec in "${specs[@]}"; do
  set -- $spec; l=$1; shift
  cp gpurun_ab/$l s-blas_amd/lib/libsblas_hip.so
  env "$@" python bench.py --ncols $n --no-extras --cpu-seconds 0 --steps 100 > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; cp /tmp/orig.so s-blas_amd/lib/libsblas_hip.so; exit 1; }
  python - <<PY
import json;d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]);print('$spec', $n, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['panels'], d['oracle_check'])
PY
done; done
cp /tmp/orig.so s-blas_amd/lib/libsblas_hip.so
