# A/B of library builds on one box through tools/spmm_shapes.py (prints times even when the result check fails: timing
# experiments): tools/ab_shapes.sh "<shape> --n N ..." lib1 lib2 ... (files under gpurun_ab/), two interleaved rounds
cd $GRAFT_REPO_ROOT
spec=$1; shift
cp s-blas_amd/lib/libsblas_hip.so /tmp/orig.so
for round in 1 2; do for l in "$@"; do
  cp gpurun_ab/$l s-blas_amd/lib/libsblas_hip.so
  python tools/spmm_shapes.py $spec --variants auto --rounds 3 --steps 100 2>&1 | grep "N=" | cut -c1-120 | sed "s/^/$l /"
done; done
cp /tmp/orig.so s-blas_amd/lib/libsblas_hip.so
