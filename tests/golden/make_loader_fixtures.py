"""Generates tests/golden/loader_cases/*.mtx and loader_expected.json.

The .mtx inputs are written by this script (they are ours); the EXPECTED CSR arrays are produced by
the reference's own loader (mmio.h + mmio_highlevel.h compiled from /root/reference into
oracle/_ref/libref_loader.so by oracle/Makefile).  Run in the build container only:
    python tests/golden/make_loader_fixtures.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle_py  # noqa: E402

CASES = {
    "general_real_unsorted": """%%MatrixMarket matrix coordinate real general
% entries deliberately not in row/column order; duplicates kept
5 4 9
3 2 -1.5
1 4 2.25e+00
5 1 7
1 1 0.125
3 2 4.0
2 3 -3e-2
4 4 1e3
1 2 .5
5 4 -0.0
""",
    "symmetric_integer": """%%MatrixMarket matrix coordinate integer symmetric
%comment
6 6 8
1 1 4
2 1 -2
4 2 7
3 3 5
6 1 9
5 4 1
6 6 3
6 5 -8
""",
    "hermitian_complex": """%%MatrixMarket matrix coordinate complex hermitian
4 4 5
1 1 2.0 0.0
2 1 1.5 -0.5
3 3 9.75 0
4 1 -4.0 2.0
4 3 0.25 1e-3
""",
    "skew_real": """%%MatrixMarket matrix coordinate real skew-symmetric
4 4 3
2 1 1.0
3 1 -2.0
4 3 3.5
""",
    "pattern_general_rect": """%%MatrixMarket matrix coordinate pattern general
3 7 6
1 7
3 1
2 4
1 2
3 6
3 2
""",
    "blank_line_before_size_mixed_case": """%%MatrixMarket MATRIX Coordinate Real General
% a blank line follows the comments

3 3 4
1 1 1.0
2 2 2.0
3 3 3.0
1 3 -1.0
""",
    "empty_rows_real": """%%MatrixMarket matrix coordinate real general
6 6 4
2 2 1.0
2 5 2.0
5 1 3.0
5 6 4.0
""",
}


def main():
    if oracle_py.ref_loader() is None:
        raise SystemExit("oracle/_ref/libref_loader.so missing: run `make -C oracle` where /root/reference exists")
    out_dir = os.path.join(HERE, "loader_cases")
    os.makedirs(out_dir, exist_ok=True)
    expected = {"_provenance": "expected arrays = output of the reference's mmio_info/mmio_data (oracle/_ref) on these files"}
    for name, text in CASES.items():
        path = os.path.join(out_dir, name + ".mtx")
        with open(path, "w") as f:
            f.write(text)
        m, n, nnz, sym, rowptr, colidx, val = oracle_py.read_mtx_ref(path)
        expected[name] = dict(m=m, n=n, nnz=nnz, symmetric=sym, rowptr=rowptr.tolist(), colidx=colidx.tolist(),
                              val=[float(v).hex() for v in val])
    with open(os.path.join(HERE, "loader_expected.json"), "w") as f:
        json.dump(expected, f, indent=1)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
