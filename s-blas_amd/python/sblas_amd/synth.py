"""Synthetic CSR stand-ins for the SuiteSparse matrices BASELINE.json names (the files are not in
the container and cannot be fetched).  Deterministic (numpy MT19937, seed 211), generated straight
into CSR, columns ascending inside a row, values uniform in [0, 1).

banded(): every row holds `nnz_per_row` nonzeros inside the band [i-half_band, i+half_band] clipped
to the matrix.  The window is cut into nnz_per_row equal strata and one column is drawn uniformly
from each, so a row's columns are distinct, sorted and spread over the whole band (about 10 % band
density for the nd24k stand-in: 399 of 4001).
"""
import numpy as np

SEED = 211


def banded(rows, nnz_per_row, half_band, seed=SEED, cols=None, chunk_rows=8192):
    cols = rows if cols is None else cols
    rng = np.random.Generator(np.random.MT19937(seed))
    i = np.arange(rows, dtype=np.int64)
    lo = np.clip(i - half_band, 0, cols - 1)
    hi = np.clip(i + half_band + 1, 1, cols)          # exclusive
    width = hi - lo
    per_row = np.minimum(nnz_per_row, width)
    rowptr = np.zeros(rows + 1, np.int64)
    np.cumsum(per_row, out=rowptr[1:])
    nnz = int(rowptr[-1])
    if nnz >= 2 ** 31:
        raise ValueError("int32 CSR overflow")
    colidx = np.empty(nnz, np.int32)
    uniform = bool((per_row == nnz_per_row).all())
    for r0 in range(0, rows, chunk_rows):
        r1 = min(rows, r0 + chunk_rows)
        if uniform:
            k = np.arange(nnz_per_row, dtype=np.int64)[None, :]
            w = width[r0:r1, None]
            b0 = (k * w) // nnz_per_row
            b1 = ((k + 1) * w) // nnz_per_row
            u = rng.random((r1 - r0, nnz_per_row))
            c = lo[r0:r1, None] + b0 + np.floor(u * (b1 - b0)).astype(np.int64)
            colidx[rowptr[r0]:rowptr[r1]] = c.reshape(-1)
        else:
            for r in range(r0, r1):
                m = int(per_row[r])
                k = np.arange(m, dtype=np.int64)
                b0 = (k * width[r]) // m
                b1 = ((k + 1) * width[r]) // m
                colidx[rowptr[r]:rowptr[r + 1]] = lo[r] + b0 + np.floor(rng.random(m) * (b1 - b0)).astype(np.int64)
    val = rng.random(nnz)
    return rowptr.astype(np.int32), colidx, val


def nd24k_like(scale=1.0):
    """M = K = 72 000, 399 nnz/row (28 728 000 nnz), band +-2000.  scale < 1 shrinks
    the row count only (same row length and band) for tests the CPU oracle must finish in seconds."""
    rows = max(64, int(round(72000 * scale)))
    return rows, banded(rows, 399, 2000)


def queen_like(rows, seed=SEED, half_band=50000, progress=None):
    """Stand-in for SuiteSparse Queen_4147 (M = 4 147 110, ~76 nnz/row, 3-D structural FEM): every row holds 20..30
    clusters of 3 consecutive columns (3 dofs per mesh node) at stencil-like offsets inside +-half_band, the diagonal
    cluster always present; ascending, duplicate-free.  `rows` scales the matrix (band is clipped to it)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    hb = min(half_band, max(3, rows // 2 - 3))
    noff = 40
    offsets = np.unique(np.concatenate([[0], (rng.integers(-hb, hb + 1, noff) // 3) * 3]))
    nclus = rng.integers(20, 31, rows)
    rowptr = np.zeros(rows + 1, np.int64)
    cols_list = []
    for r0 in range(0, rows, 65536):
        r1 = min(rows, r0 + 65536)
        chunk = []
        for r in range(r0, r1):
            pick = np.sort(rng.choice(len(offsets), min(nclus[r], len(offsets)), replace=False))
            base = (r // 3) * 3 + offsets[pick]
            base = base[(base >= 0) & (base + 2 < rows)]
            if base.size == 0:
                base = np.array([min(max((r // 3) * 3, 0), rows - 3)])
            c = (base[:, None] + np.arange(3)[None, :]).reshape(-1)
            chunk.append(np.unique(c))
            rowptr[r + 1] = chunk[-1].size
        cols_list.extend(chunk)
        if progress is not None:
            progress(r1)
    np.cumsum(rowptr, out=rowptr)
    colidx = np.concatenate(cols_list).astype(np.int32)
    val = rng.random(colidx.size) * 2.0 - 1.0
    return rowptr.astype(np.int32), colidx, val


def random_csr(rows, cols, avg_nnz, seed=SEED, sorted_rows=False, empty_every=0, long_row=None):
    """Unstructured test matrix: duplicate columns allowed, rows unsorted unless asked, optional
    empty rows (every `empty_every`-th) and one long row (index, length)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    lens = rng.poisson(avg_nnz, rows).astype(np.int64)
    if empty_every:
        lens[::empty_every] = 0
    if long_row is not None:
        lens[long_row[0]] = long_row[1]
    rowptr = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = rng.integers(0, max(cols, 1), nnz).astype(np.int32)
    if sorted_rows:
        for r in range(rows):
            colidx[rowptr[r]:rowptr[r + 1]].sort()
    val = rng.random(nnz) * 2.0 - 1.0
    return rowptr.astype(np.int32), colidx, val
