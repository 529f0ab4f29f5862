cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "spmv" 2>&1 | tail -2
python tools/spmv_shapes.py banded:2000000:3:20 banded:2000000:5:20 banded:1000000:7:20 powerlaw:1000000 --variants auto,stream,plain --rounds 2 --steps 20 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-110
python tools/spmv_shapes.py banded:600000:48:2000 banded:600000:56:2000 banded:600000:64:2000 banded:600000:72:2000 banded:600000:90:2000 queen:300000 --variants auto,stream,seg4 --rounds 2 --steps 20 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-110
