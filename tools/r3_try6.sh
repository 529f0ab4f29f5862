set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "long_rows or power_law" > gpurun_out/r3_long.txt 2>&1 || (tail -30 gpurun_out/r3_long.txt; exit 1)
tail -2 gpurun_out/r3_long.txt
for spec in "powerlaw:1000000:40:100000 128" "powerlaw:1000000:40:1000000 128" "powerlaw:1000000:40:1000000 256" "powerlaw:300000:40:300000 64"; do
set -- $spec
python tools/spmm_shapes.py $1 --n $2 --variants "auto,auto=SBLAS_TUNE=0:0:1073741824:0" --rounds 2 --steps 5
done > gpurun_out/r3_longrow_shapes.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3_longrow_shapes.txt
