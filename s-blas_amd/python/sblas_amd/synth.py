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


def block_structured(rows, nnz_per_row=399, half_band=2000, block_rows=16, block_cols=4, fill=0.6, seed=SEED):
    """nd24k-like rows (same row length and band as nd24k_like) whose nonzeros sit in dense sub-blocks, the shape
    supernodal / multi-dof FEM matrices have and the uniform stand-in lacks: the matrix is cut into block_rows x
    block_cols aligned blocks, every block row (16 matrix rows) picks the block columns it uses inside its band, and
    every (row, column) position of a picked block holds a nonzero with probability `fill`.  The number of picked
    blocks is chosen so that rows average nnz_per_row.  Columns ascending per row, values U[0, 1)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    nbr = (rows + block_rows - 1) // block_rows
    per_blockrow = max(1, int(round(nnz_per_row / (block_cols * fill))))
    counts = np.zeros(rows, np.int64)
    cols_list = []
    for br in range(nbr):
        r0, r1 = br * block_rows, min(rows, (br + 1) * block_rows)
        lo = max(0, r0 - half_band) // block_cols
        hi = min(rows, r1 + half_band) // block_cols            # exclusive block-column bound
        picked = np.sort(rng.choice(np.arange(lo, max(hi, lo + 1)), min(per_blockrow, max(hi - lo, 1)), replace=False))
        # positions: (r1 - r0) x (picked * block_cols + 0..block_cols-1), kept with probability `fill`
        colgrid = (picked[:, None] * block_cols + np.arange(block_cols)[None, :]).reshape(-1)
        colgrid = colgrid[colgrid < rows]
        keep = rng.random((r1 - r0, colgrid.size)) < fill
        for i in range(r1 - r0):
            c = colgrid[keep[i]]
            if c.size == 0:
                c = colgrid[:1]
            cols_list.append(c.astype(np.int32))
            counts[r0 + i] = c.size
    rowptr = np.zeros(rows + 1, np.int64)
    np.cumsum(counts, out=rowptr[1:])
    colidx = np.concatenate(cols_list)
    val = rng.random(colidx.size)
    return rowptr.astype(np.int32), colidx, val


def queen_like_grid(rows, seed=SEED, half_band=50000, keep=0.93, dofs=3):
    """Queen_4147-like rows with the locality a 3-D FEM numbering has (queen_like() scatters its 40 offsets uniformly,
    which no mesh ordering does): nodes of a structured nx x ny x nz grid numbered x-fastest, 3 dofs per node, every
    node coupled to its 27 grid neighbours (each kept with probability `keep`, the diagonal always), i.e. up to
    27 x 3 = 81 columns per row in nine runs of nine consecutive columns; the plane size nx*ny is chosen so that
    the farthest neighbour sits `half_band` rows away (3 * (nx*ny + nx + 1) ~ half_band).  `rows` is rounded down
    to whole nodes.  Columns ascending, duplicate-free, values U[-1, 1).  `dofs`: unknowns per node (3 = Queen_4147's;
    2 and 6 exercise the row-merging kernel's group detection)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    nodes = max(rows // dofs, 8)
    plane = max(4, min(half_band // dofs, nodes // 3))
    nx = max(2, int(np.sqrt(plane)))
    ny = max(2, plane // nx)
    plane = nx * ny
    nz = max(1, (nodes + plane - 1) // plane)
    nodes = min(nodes, plane * nz)
    rows = nodes * dofs
    n = np.arange(nodes, dtype=np.int64)
    x, y, z = n % nx, (n // nx) % ny, n // plane
    cols_parts, cnt = [], np.zeros(nodes, np.int64)
    offs = [(dx, dy, dz) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    nb = np.empty((nodes, 27), np.int64)
    ok = np.empty((nodes, 27), bool)
    for k, (dx, dy, dz) in enumerate(offs):
        xx, yy, zz = x + dx, y + dy, z + dz
        m = n + dx + nx * dy + plane * dz
        ok[:, k] = (xx >= 0) & (xx < nx) & (yy >= 0) & (yy < ny) & (zz >= 0) & (m >= 0) & (m < nodes)
        nb[:, k] = m
    drop = rng.random((nodes, 27)) >= keep
    drop[:, 13] = False                                           # the node itself stays
    ok &= ~drop
    per_node = ok.sum(1)
    # rows of a node share its column set: 3 dofs x (neighbours x 3 dofs)
    rowptr = np.zeros(rows + 1, np.int64)
    np.cumsum(np.repeat(per_node * dofs, dofs), out=rowptr[1:])
    colidx = np.empty(int(rowptr[-1]), np.int32)
    flat_nb = nb[ok]                                              # neighbours in ascending order per node (offs is sorted)
    node_start = np.zeros(nodes + 1, np.int64)
    np.cumsum(per_node, out=node_start[1:])
    c3 = (flat_nb[:, None] * dofs + np.arange(dofs)[None, :]).reshape(-1)   # columns of one row of each node, node after node
    # scatter: row 3n+d gets the 3*per_node[n] columns of node n
    starts = rowptr[:-1].reshape(nodes, dofs)
    lens = per_node * dofs
    src_start = node_start[:-1] * dofs
    idx_in = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(src_start, lens)
    for d in range(dofs):
        colidx[np.repeat(starts[:, d], lens) + idx_in] = c3
    val = rng.random(colidx.size) * 2.0 - 1.0
    return rowptr.astype(np.int32), colidx, val


def powerlaw(rows, avg=3.0, max_len=5000, alpha=2.25, seed=SEED, cols=None):
    """webbase-1M-like (the reference authors' own SpMV profiling input, profiling.sh:16,21): row lengths from a
    truncated power law (most rows hold 1-3 nonzeros, a few hold thousands; mean ~avg, maximum max_len), columns uniform
    over the matrix, ascending and duplicate-free inside a row, values U[-1, 1)."""
    cols = rows if cols is None else cols
    rng = np.random.Generator(np.random.MT19937(seed))
    u = rng.random(rows)
    # inverse CDF of a Pareto tail on [1, max_len]
    a = alpha - 1.0
    lens = np.floor((1.0 - u * (1.0 - max_len ** (-a))) ** (-1.0 / a)).astype(np.int64)
    lens = np.clip(lens, 1, min(max_len, cols))
    for _ in range(4):                                            # scale the body towards the requested mean, keep the tail
        scale = avg / lens.mean()
        if abs(scale - 1.0) < 0.02:
            break
        lens = np.clip(np.round(lens * np.where(lens < 64, scale, 1.0)).astype(np.int64), 1, min(max_len, cols))
    lens[rng.integers(0, rows)] = min(max_len, cols)              # the maximum is always present
    rowptr = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = rng.integers(0, cols, nnz).astype(np.int64)
    # sort inside rows: key = row * cols + col
    key = np.repeat(np.arange(rows, dtype=np.int64), lens) * cols + colidx
    key.sort()
    colidx = (key % cols).astype(np.int32)
    # duplicates inside a row are legal CSR (the reference adds them up); leave them in
    val = rng.random(nnz) * 2.0 - 1.0
    return rowptr.astype(np.int32), colidx, val
