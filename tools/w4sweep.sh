#!/bin/bash
# A/B of pre-built libsblas_hip.so variants (tools/w4libs/lib_<RPW>_<NBUF>.so) of the gen-4 kernel
cp s-blas_amd/lib/libsblas_hip.so /tmp/lib_orig.so
for f in tools/w4libs/lib_*.so; do
  cp $f s-blas_amd/lib/libsblas_hip.so
  SBLAS_SPMM_VARIANT=win4 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('$f kernel=%.4f ms check=%s' % (r['kernel_ms'], d['oracle_check']))"
done
cp /tmp/lib_orig.so s-blas_amd/lib/libsblas_hip.so
