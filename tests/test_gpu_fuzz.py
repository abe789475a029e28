"""Seeded random shapes through every kernel and layout of the engine, against the oracle (`pytest -m gpu`).

The fixed cases of test_gpu_parity.py are chosen by hand; these are not: row counts from 1 to a few hundred thousand,
column counts from 1 to millions, row lengths constant / Poisson / power-law with empty rows, columns uniform / banded /
clustered, every CSR kernel (row-parallel, one lane per row, LDS window where it fits, panel in each of its layouts,
chunk sizes, pipelines and wavefront syncs, two-phase with several panel widths), ELL handles from clean stencils to
noise, COO in file order with duplicates.  Gate: SURVEY 8d's two tolerances for kernels that reorder sums, equality bit
for bit with the fma oracle for the kernels that keep the reference's order (ELL, one lane per row)."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
BASE = int(os.environ.get("SPMV_FUZZ_BASE", "0"))  # other seeds: SPMV_FUZZ_BASE=1000 pytest -m gpu tests/test_gpu_fuzz.py


def _random_csr(rng):
    nrow = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 20_001, 60_000, 250_000]))
    ncol = int(rng.choice([1, 7, 64, 1000, 20_000, 131_073, 1_000_000, 3_000_000]))
    kind = rng.choice(["const", "poisson", "power", "mostly_empty"])
    if kind == "const":
        lens = np.full(nrow, int(rng.integers(1, 40)))
    elif kind == "poisson":
        lens = rng.poisson(rng.uniform(0.5, 24), nrow)
    elif kind == "power":
        lens = np.minimum(3000, (4.0 / rng.uniform(1e-4, 1, nrow)).astype(np.int64))
    else:
        lens = np.where(rng.uniform(size=nrow) < 0.9, 0, rng.integers(1, 200, nrow))
    lens = np.minimum(lens, 50_000_000 // max(nrow, 1)).astype(np.int64)
    rp = np.zeros(nrow + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    rows = np.repeat(np.arange(nrow, dtype=np.int64), lens)
    cols_kind = rng.choice(["uniform", "band", "clustered"])
    if cols_kind == "uniform" or ncol < 16:
        cc = rng.integers(0, ncol, nnz)
    elif cols_kind == "band":
        w = int(rng.choice([3, 64, 5000]))
        cc = (rows * ncol // max(nrow, 1) + rng.integers(-w, w + 1, nnz)) % ncol
    else:
        centres = rng.integers(0, ncol, 5)
        cc = (centres[rng.integers(0, 5, nnz)] + rng.integers(0, 300, nnz)) % ncol
    return nrow, ncol, rp.astype(np.int32), cc.astype(np.int32), rng.uniform(-1, 1, nnz)


@pytest.mark.parametrize("seed", range(40))
def test_random_csr_shapes_through_every_kernel(ctx, orc, pkg, seed):
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 1000 + seed)
    nrow, ncol, rp, cc, cv = _random_csr(rng)
    nnz = len(cv)
    x = rng.uniform(-1, 1, ncol)
    ref, ref_fma, scale = np.zeros(nrow), np.zeros(nrow), np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_spmv(orc, rp, cc, cv, x, ref_fma, fma=True)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    what = f"seed {seed}: {nrow} x {ncol}, {nnz} entries"

    def run(A, label, exact=False):
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.apply(A, dx, dy)  # accumulates
        ctx.sync()
        got = dy.download()
        if exact:  # y0 + s then + s again, s the row sum in the reference's order with fma
            assert np.array_equal(got, ref_fma + ref_fma), f"{what} {label}"
        else:
            ol.assert_parity(got, 2 * ref, scale, f"{what} {label}", reps=2)

    A = ctx.csr(nrow, ncol, rp, cc, cv)
    run(A, f"auto (kernel {A.info.kernel})")
    for lanes in (1, 4, 32):
        A.set_kernel(capi.CSR_VECTOR, lanes)
        run(A, f"vector lanes={lanes}")
    A.set_kernel(capi.CSR_SCALAR)
    run(A, "one lane per row", exact=True)
    if A.get_param("window_max_span") and A.get_param("window_max_span") <= 8192:
        A.set_kernel(capi.CSR_LDSWIN, 8)
        run(A, "lds window")
    if nnz:
        combos = [(int(rng.choice([0, 3, 4])), int(rng.choice([2, 4, 8, 16])), int(rng.choice([0, 1, 2])), int(rng.choice([0, 1, 3])),
                   int(rng.choice([0, 0, 7, 333, 20_000])), int(rng.choice([0, 16, 4096]))) for _ in range(4)]
        combos.append((4, 8, 2, 1, 0, 0))  # the C2 instance
        for layout, unroll, pipe, sync, rows, width in combos:
            for k, v in (("panel_aos", layout), ("panel_unroll", unroll), ("panel_pipe", pipe), ("panel_sync", sync), ("panel_rows", rows),
                         ("panel_width", width)):
                A.set_param(k, v)
            A.set_kernel(capi.CSR_PANEL)
            run(A, f"panel layout={layout}->{A.get_param('panel_layout')} unroll={unroll} pipe={pipe} sync={sync} rows={rows} width={width}")
        if nnz + 16 * ((ncol + 6999) // 7000) * 256 < 2**31:
            for cols, rotate in ((20_000, 256), (7_000, 0), (20_000, 8)):
                A.set_param("twophase_panel_cols", cols)
                A.set_param("twophase_rotate", rotate)
                A.set_kernel(capi.CSR_TWOPHASE)
                run(A, f"two-phase cols={cols} rotate={rotate}")


@pytest.mark.parametrize("seed", range(24))
def test_random_ell_handles_from_stencils_to_noise(ctx, orc, pkg, seed):
    """column-major ELL with a random mix of diagonal slots, padding and arbitrary columns: the diagonal-slot kernel (when
    the analysis takes it) and the plain ones must equal the fma oracle exactly"""
    rng = np.random.default_rng(BASE + 2000 + seed)
    nrow = int(rng.choice([1024, 1026, 5000, 33_334, 120_000]))
    ncol = int(rng.choice([nrow, nrow + 77, max(64, nrow // 3), 4 * nrow]))
    k = int(rng.integers(1, 12))
    far = int(rng.choice([3, 200, 9000]))
    offs = np.sort(rng.choice(np.arange(-far, far + 1), size=k, replace=k > 2 * far + 1))
    rows = np.arange(nrow)
    col = rows[None, :] * ncol // nrow + offs[:, None] if rng.uniform() < 0.3 else rows[None, :] + offs[:, None]
    val = rng.uniform(-1, 1, (k, nrow))
    if rng.uniform() < 0.5:
        col = col % ncol
    else:
        out = (col < 0) | (col >= ncol)
        col[out], val[out] = 0, 0.0
    noise = rng.choice([0.0, 0.001, 0.2, 0.8])
    bad = rng.uniform(size=(k, nrow)) < noise
    col[bad] = rng.integers(0, ncol, int(bad.sum()))
    col, val = col.astype(np.int32).ravel(), val.ravel()
    x = rng.uniform(-1, 1, ncol)
    ref = np.zeros(nrow)
    ol.ell_spmv(orc, nrow, k, col, val, x, ref, fma=True)
    A = ctx.ell(nrow, ncol, k, nrow * k, col, val)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    # AUTO (round 5: the format's own variants and, where the columns are scattered, a row-grouped copy are timed): whichever
    # stayed, the product is within the parity gate; the format's own kernels are bit-identical to the fma oracle
    scale = np.zeros(nrow)
    ol.ell_spmv(orc, nrow, k, np.ascontiguousarray(col), np.abs(val), np.abs(x), scale)
    dy.fill(0.0)
    ctx.apply(A, dx, dy)
    ctx.sync()
    what = f"seed {seed}: ELL {nrow} x {ncol}, k = {k}, AUTO (kernel {A.info.kernel}, variant {A.get_param('ell_variant')}, copy runs {A.get_param('rowgrouped_kernel')})"
    ol.assert_parity(dy.download(), ref, np.maximum(scale, 1e-300), what)
    if A.info.kernel == pkg.capi.CSR_VECTOR:
        assert np.array_equal(dy.download(), ref), what
    for lanes in (2, 1):  # two rows per lane (diagonal slots where found; flags 8: every index read), one row per lane
        A.set_kernel(pkg.capi.CSR_VECTOR, lanes)
        for flags in (0, 8):
            A.set_flags(flags)
            dy.fill(0.0)
            ctx.apply(A, dx, dy)
            ctx.sync()
            assert np.array_equal(dy.download(), ref), (seed, nrow, ncol, k, lanes, flags, A.get_param("ell_diagonal_slots"))


@pytest.mark.parametrize("seed", range(16))
def test_random_coo_in_file_order_with_duplicates(ctx, orc, pkg, seed):
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 3000 + seed)
    nrow = int(rng.choice([1, 17, 3000, 90_000]))
    ncol = int(rng.choice([1, 100, 50_000, 2_000_000]))
    nnz = int(rng.choice([0, 1, 1000, 400_000, 3_000_000]))
    row = rng.integers(0, nrow, nnz).astype(np.int32)
    if rng.uniform() < 0.5:
        row = np.sort(row)
    if nnz > 10 and rng.uniform() < 0.5:
        row[: nnz // 3] = row[0]  # a hub row
    col = rng.integers(0, ncol, nnz).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    x = rng.uniform(-1, 1, ncol)
    ref = np.zeros(nrow)
    ol.coo_spmv(orc, row, col, val, x, ref)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    scale = np.zeros(nrow)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.coo(nrow, ncol, row, col, val)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    for kernel in (capi.CSR_AUTO, capi.CSR_VECTOR, capi.CSR_PANEL):
        A.set_kernel(kernel)
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, f"coo seed {seed}: {nrow} x {ncol}, {nnz} entries, kernel {kernel}")
    # the segmented scan over a copy of the entries in column bins (made unasked only for a large handle with a large x)
    A.set_kernel(capi.CSR_VECTOR)
    per_xcd = int(rng.integers(1, 9))
    A.set_param("coo_column_bins", per_xcd)
    assert A.get_param("coo_column_bins") == (8 * per_xcd if nnz else 0)
    dy.fill(0.0)
    ctx.apply(A, dx, dy)
    ctx.sync()
    ol.assert_parity(dy.download(), ref, scale, f"coo seed {seed}: {nrow} x {ncol}, {nnz} entries, scan over {8 * per_xcd} column bins")
    A.set_param("coo_column_bins", 0)
    C = ctx.coo_to_csr(A)
    assert C.info.nnz == nnz
    got = C.download()
    assert np.array_equal(got[0], rp) and np.array_equal(got[1], cc) and np.array_equal(got[2], cv)


@pytest.mark.parametrize("seed", range(16))
def test_random_dia_handles(ctx, orc, pkg, seed):
    """row-major DIA with random offsets (narrow bands: x through LDS; far offsets: x from global memory), odd and even
    diagonal counts, rectangular shapes: bit for bit the fma oracle (x padded with zeros where the reference over-reads)"""
    rng = np.random.default_rng(BASE + 4000 + seed)
    nrow = int(rng.choice([1, 255, 256, 257, 5000, 70_001]))
    ncol = int(rng.choice([nrow, max(1, nrow // 2), nrow + 300, 3 * nrow]))
    nd = int(rng.integers(1, 20))
    span = int(rng.choice([2, 40, 900, max(2, nrow)]))
    offs = np.sort(rng.choice(np.arange(-span, span + 1), size=min(nd, 2 * span + 1), replace=False)).astype(np.int32)
    val = rng.uniform(-1, 1, nrow * len(offs))
    x = rng.uniform(-1, 1, ncol)
    xpad = np.zeros(max(nrow, ncol) + 1)
    xpad[:ncol] = x
    y0 = rng.uniform(-1, 1, nrow)
    ref = y0.copy()
    ol.dia_spmv(orc, nrow, offs, val, xpad, ref, fma=True)
    A = ctx.dia(nrow, ncol, offs, val)
    dx = ctx.vector_from(x)
    for flags in (0, 4):  # 4 = SPMV_FLAG_DIA_GLOBAL_X
        A.set_flags(flags)
        dy = ctx.vector_from(y0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        assert np.array_equal(dy.download(), ref), (seed, nrow, ncol, list(offs), flags)


@pytest.mark.parametrize("seed", range(10))
def test_random_csc_handles(ctx, orc, pkg, seed):
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 5000 + seed)
    nrow = int(rng.choice([1, 300, 40_000, 400_000]))
    ncol = int(rng.choice([1, 77, 30_000, 500_000]))
    nnz = int(rng.choice([0, 5, 20_000, 2_500_000]))
    row = rng.integers(0, nrow, nnz).astype(np.int32)
    col = np.sort(rng.integers(0, ncol, nnz)).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    cp, cr, cw = ol.coo_to_csc(orc, ncol, row, col, val)
    x = rng.uniform(-1, 1, ncol)
    ref = np.zeros(nrow)
    ol.csc_spmv(orc, cp, cr, cw, x, ref)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    scale = np.zeros(nrow)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.csc(nrow, ncol, cp, cr, cw)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    for kernel in (capi.CSR_AUTO, capi.CSR_VECTOR):
        A.set_kernel(kernel)
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, f"csc seed {seed}: {nrow} x {ncol}, {nnz} entries, kernel {kernel}")


@pytest.mark.parametrize("seed", range(12))
def test_random_systems_through_the_gauss_seidel_sweep(ctx, orc, pkg, seed):
    """random diagonally dominant matrices — symmetric or not in pattern, rows in random order, the diagonal entry
    sometimes split in two — in both sweep orders, 1-3 sweeps, against the oracle's sweep over the sequence the engine
    reports; the multicolour sequence against the oracle's sequential greedy colouring"""
    rng = np.random.default_rng(BASE + 6000 + seed)
    n = int(rng.choice([2, 65, 1000, 20_000, 150_000]))
    k = int(rng.integers(0, 9))
    cols = rng.integers(0, n, (n, k))
    if rng.uniform() < 0.5 and k:  # local couplings: long dependency chains in row order
        cols = (np.arange(n)[:, None] + rng.integers(-3, 4, (n, k))) % n
    vals = rng.uniform(-1, 1, (n, k))
    rows = np.repeat(np.arange(n), k).reshape(n, k)
    vals[cols == rows] = 0.0
    r, c, v = rows.ravel(), cols.ravel(), vals.ravel()
    if rng.uniform() < 0.5:  # symmetric pattern
        r, c, v = np.concatenate([r, c]), np.concatenate([c, r]), np.concatenate([v, v])
    dom = np.zeros(n)
    np.add.at(dom, r, np.abs(v))
    dom += 1.0
    split = rng.uniform() < 0.5
    dr = np.arange(n)
    r = np.concatenate([r, dr] + ([dr] if split else []))
    c = np.concatenate([c, dr] + ([dr] if split else []))
    v = np.concatenate([v, dom * (0.6 if split else 1.0)] + ([dom * 0.4] if split else []))
    o = np.lexsort((rng.uniform(size=len(r)), r))  # rows together, entries inside a row in random order
    r, c, v = r[o], c[o], v[o]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp, cc = np.cumsum(rp).astype(np.int32), c.astype(np.int32)
    A = ctx.csr(n, n, rp, cc, v)
    b_host, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b = ctx.vector_from(b_host)
    for order in (1, 0):
        A.set_param("symgs_order", order)
        seq = ctx.symgs_order(A)
        if order == 1:
            ncol, _, want_seq = ol.greedy_colour_order(orc, rp, cc)
            assert np.array_equal(seq, want_seq) and A.get_param("symgs_colours") == ncol
        else:
            assert np.array_equal(seq, np.arange(n))
        sweeps = int(rng.integers(1, 4))
        want = x0.copy()
        assert ol.symgs(orc, rp, cc, v, b_host, want, sweeps, order=seq) == 0
        x = ctx.vector_from(x0)
        ctx.symgs(A, b, x, sweeps)
        ctx.sync()
        err = np.max(np.abs(x.download() - want)) / max(np.max(np.abs(want)), 1e-300)
        assert err <= ol.REL_TOL, (seed, n, k, order, sweeps, err)


@pytest.mark.parametrize("seed", range(10))
def test_random_blas1(ctx, orc, seed):
    """dot and the axpby branches on random lengths and coefficients (0, 1, -1 and general), w aliasing x or y"""
    rng = np.random.default_rng(BASE + 7000 + seed)
    n = int(rng.choice([1, 2, 3, 255, 1025, 100_003, 3_000_001]))
    xh, yh = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    x, y = ctx.vector_from(xh), ctx.vector_from(yh)
    d = ctx.dot(x, y)
    assert abs(d - float(xh @ yh)) <= 1e-12 * float(np.abs(xh) @ np.abs(yh)) + 1e-300
    for alpha in (0.0, 1.0, -1.0, float(rng.uniform(-2, 2))):
        for beta in (0.0, 1.0, -1.0, float(rng.uniform(-2, 2))):
            want = np.zeros(n)
            ol.axpby(orc, alpha, xh, beta, yh, want, fma=True)
            w = ctx.vector(n)
            w.fill(np.nan)
            ctx.axpby(alpha, x, beta, y, w)
            ctx.sync()
            assert np.array_equal(w.download(), want), (n, alpha, beta)
    xa = ctx.vector_from(xh)  # w = x
    ctx.axpby(0.5, xa, 2.0, y, xa)
    want = np.zeros(n)
    ol.axpby(orc, 0.5, xh, 2.0, yh, want, fma=True)
    ctx.sync()
    assert np.array_equal(xa.download(), want)


@pytest.mark.parametrize("seed", range(12))
def test_random_conversions_shards_and_column_splits(ctx, orc, pkg, seed):
    """COO -> CSR / ELL on the device against the oracle's arrays (bit-exact, in-row order kept); row shards of the CSR
    (the reference's equal-rows partition, src/mat_vec.cpp:245-246) concatenated; the split by a column range: inside
    + outside = the whole product, inside rebased"""
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 8000 + seed)
    nrow = int(rng.choice([1, 9, 500, 30_000, 200_000]))
    ncol = int(rng.choice([1, 40, 9_000, 700_000]))
    nnz = int(rng.choice([0, 3, 5_000, 900_000]))
    row = rng.integers(0, nrow, nnz).astype(np.int32)
    if rng.uniform() < 0.5:
        row = np.sort(row)
    col = rng.integers(0, ncol, nnz).astype(np.int32)
    val = rng.uniform(-1, 1, nnz)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    A = ctx.coo(nrow, ncol, row, col, val)
    Cm = ctx.coo_to_csr(A)
    a, b, v = Cm.download()
    assert np.array_equal(a, rp) and np.array_equal(b, cc) and np.array_equal(v, cv)
    if nrow * max(1, int(np.diff(rp).max())) <= 8_000_000:
        k, ec, ev = ol.coo_to_ell(orc, nrow, row, col, val)
        for E in (ctx.coo_to_ell(A), ctx.csr_to_ell(Cm)):
            assert E.info.ell_k == k
            _, gc, gv = E.download()
            assert np.array_equal(gc, ec) and np.array_equal(gv, ev)
    x = rng.uniform(-1, 1, ncol)
    ref, scale = np.zeros(nrow), np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx = ctx.vector_from(x)
    # row shards
    nparts = int(rng.choice([1, 2, 3, 8, 17]))
    rp64 = rp.astype(np.int64)
    got = np.zeros(nrow)
    for part in range(nparts):
        r0, r1 = ol.partition_rows(orc, nrow, nparts, part)
        S = ctx.csr_shard(r0, r1, ncol, rp64, cc, cv)
        assert S.info.nrow == r1 - r0 and S.info.row_begin == r0
        y = ctx.vector(r1 - r0)
        y.fill(0.0)
        ctx.apply(S, dx, y)
        ctx.sync()
        got[r0:r1] = y.download()
    ol.assert_parity(got, ref, scale, f"seed {seed}: {nparts} row shards")
    # column split
    c0 = int(rng.integers(0, ncol))
    c1 = int(rng.integers(c0, ncol + 1))
    inside, outside = ctx.csr_split_columns(Cm, c0, c1)
    assert inside.info.ncol == c1 - c0 and inside.info.nnz + outside.info.nnz == nnz
    y = ctx.vector(nrow)
    y.fill(0.0)
    ctx.apply(inside, ctx.vector_from(x[c0:c1]), y)
    ctx.apply(outside, dx, y)
    ctx.sync()
    ol.assert_parity(y.download(), ref, scale, f"seed {seed}: columns [{c0}, {c1}) split off")


@pytest.mark.parametrize("seed", range(6))
def test_random_large_csr_shapes(ctx, orc, pkg, seed):
    """a few million rows: whole rounds of row groups, thousands of slices, ragged chunk tails, thousands of two-phase
    panels; ragged row lengths; square, wide and tall"""
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 9000 + seed)
    nrow = int(rng.choice([1_000_003, 2_500_000, 4_194_304]))
    ncol = int(rng.choice([300_000, nrow, 31_000_000]))
    lens = rng.poisson(rng.uniform(3, 14), nrow).astype(np.int64)
    lens[rng.integers(0, nrow, 5)] = 3000  # a few long rows
    rp = np.zeros(nrow + 1, np.int64)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    cc = rng.integers(0, ncol, nnz).astype(np.int32)
    cv = rng.uniform(-1, 1, nnz)
    rp = rp.astype(np.int32)
    x = rng.uniform(-1, 1, ncol)
    ref, scale = np.zeros(nrow), np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.csr(nrow, ncol, rp, cc, cv)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    runs = [("auto", lambda: None), ("vector", lambda: A.set_kernel(capi.CSR_VECTOR)), ("panel", lambda: A.set_kernel(capi.CSR_PANEL)),
            ("two-phase", lambda: A.set_kernel(capi.CSR_TWOPHASE)), ("scan", lambda: A.set_kernel(capi.CSR_SEGSCAN)),
            ("split, default threshold", lambda: A.set_kernel(capi.CSR_SPLIT)),
            ("split, rows of 3 and more are long", lambda: (A.set_param("split_row_threshold", 3), A.set_kernel(capi.CSR_SPLIT))),
            ("split, every row is long", lambda: (A.set_param("split_row_threshold", 1), A.set_kernel(capi.CSR_SPLIT))),
            ("split, rows of 3 and more as virtual rows", lambda: (A.set_param("split_row_threshold", 3), A.set_param("split_mode", 2), A.set_kernel(capi.CSR_SPLIT))),
            ("split, every row as virtual rows", lambda: (A.set_param("split_row_threshold", 1), A.set_kernel(capi.CSR_SPLIT))),
            ("split, rows of 3 and more in chunks", lambda: (A.set_param("split_row_threshold", 3), A.set_param("split_mode", 1), A.set_kernel(capi.CSR_SPLIT)))]
    def ell_copy():
        try:
            A.set_kernel(capi.CSR_ELL)
        except capi.SpmvError as e:  # (refused where the padding to the longest row would be beyond 16x the entries, or a row is empty)
            assert "out of proportion" in str(e) or "empty row" in str(e), e
            A.set_kernel(capi.CSR_VECTOR)

    runs.append(("ell copy", ell_copy))
    for name, setup in runs:
        setup()
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, f"seed {seed}: {nrow} x {ncol}, {nnz} entries, {name} (kernel {A.info.kernel})")
        if seed % 2 == 0:
            # round 6: whatever state the handle is in, its plan rebuilds it on another handle of the same matrix - same kernel, same
            # copies, nothing timed - and the product stays within the gate
            plan = A.get_plan()
            B = ctx.csr(nrow, ncol, rp, cc, cv)
            B.set_plan(plan)
            assert B.get_plan() == plan and B.info.kernel == A.info.kernel and B.get_param("select_candidates") == 0, name
            dy.fill(0.0)
            ctx.apply(B, dx, dy)
            ctx.sync()
            ol.assert_parity(dy.download(), ref, scale, f"seed {seed}: {nrow} x {ncol}, {nnz} entries, {name} from its plan (kernel {B.info.kernel})")
            del B


@pytest.mark.parametrize("seed", range(10))
def test_random_spd_systems_through_cg_and_the_fused_dot(ctx, orc, pkg, seed):
    """random symmetric strictly diagonally dominant systems: apply_dot (overwrite or accumulate) through every CSR
    kernel equals the product and the dot of its result; CG — plain, Jacobi, Gauss-Seidel in both orders — ends with
    the residual it reports, checked through the oracle's product"""
    capi = pkg.capi
    rng = np.random.default_rng(BASE + 10_000 + seed)
    n = int(rng.choice([3, 200, 5_000, 120_000, 700_000]))
    k = int(rng.integers(1, 7))
    r = np.repeat(np.arange(n), k)
    c = rng.integers(0, n, n * k) if rng.uniform() < 0.5 else (r + rng.integers(-50, 51, n * k)) % n
    v = rng.uniform(-1, 1, n * k)
    keep = r != c
    r, c, v = r[keep], c[keep], v[keep]
    rr, cc_, vv = np.concatenate([r, c]), np.concatenate([c, r]), np.concatenate([v, v])
    diag = np.zeros(n)
    np.add.at(diag, rr, np.abs(vv))
    rr, cc_, vv = np.concatenate([rr, np.arange(n)]), np.concatenate([cc_, np.arange(n)]), np.concatenate([vv, diag + rng.uniform(0.05, 1.0, n)])
    o = np.lexsort((cc_, rr))
    rr, cc_, vv = rr[o], cc_[o], vv[o]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rr + 1, 1)
    rp, cc = np.cumsum(rp).astype(np.int32), cc_.astype(np.int32)
    A = ctx.csr(n, n, rp, cc, vv)
    xh, wh = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, vv, xh, ref)
    ol.csr_abs_row_sums(orc, rp, cc, vv, xh, scale)
    x, w, y = ctx.vector_from(xh), ctx.vector_from(wh), ctx.vector(n)
    A.set_param("split_row_threshold", 4)  # (kernel SPLIT below: rows of 4 entries and more on their own, the others through the copy)
    A.set_param("split_mode", 1 + seed % 2)  # ... in chunks / as virtual rows
    for kernel in (capi.CSR_AUTO, capi.CSR_VECTOR, capi.CSR_SCALAR, capi.CSR_PANEL, capi.CSR_TWOPHASE, capi.CSR_SEGSCAN, capi.CSR_SPLIT):
        if kernel == capi.CSR_TWOPHASE and len(vv) < 1000:
            continue
        A.set_kernel(kernel)
        for overwrite in (True, False):
            y.fill(0.25 if overwrite else 0.0)
            d = ctx.apply_dot(A, x, y, w, overwrite=overwrite)
            got = y.download()
            ol.assert_parity(got, ref, scale, f"seed {seed} n={n} kernel {kernel} overwrite={overwrite}")
            assert abs(d - float(wh @ got)) <= 1e-11 * float(np.abs(wh) @ np.abs(got)) + 1e-300
    A.set_kernel(capi.CSR_AUTO)
    bh = rng.uniform(-1, 1, n)
    b, sol = ctx.vector_from(bh), ctx.vector(n)
    for kw in ({}, {"jacobi": True}, {"symgs": True}, {"symgs": True, "row_order": True}):
        row_order = kw.pop("row_order", False)
        A.set_param("symgs_order", 0 if row_order else 1)
        sol.fill(0.0)
        iters, relres = ctx.cg(A, b, sol, max_iter=500, rel_tol=1e-9, check_every=int(rng.integers(1, 9)), **kw)
        ax = np.zeros(n)
        ol.csr_spmv(orc, rp, cc, vv, sol.download(), ax)
        true_res = np.linalg.norm(bh - ax) / np.linalg.norm(bh)
        assert relres <= 1e-9 and true_res <= 1e-7, (seed, n, kw, row_order, iters, relres, true_res)
