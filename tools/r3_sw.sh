cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r3_t.txt 2>&1; tail -5 gpurun_out/r3_t.txt
for spec in "blocks:72000:0.6 128" "blocks:72000:0.8 256" "blocks:72000:0.7 256" "qgrid:300000 256" "qgrid:300000:6 256"; do
set -- $spec
python tools/spmm_shapes.py $1 --n $2 --variants "auto" --rounds 2 --steps 5 2>&1 | grep -E "N=" | cut -c1-230 | sed "s/^/$1 /"
done
python tools/fuzz_parity.py --cases 600 2>&1 | tail -3
