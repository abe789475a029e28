"""Seeded random shapes through every kernel and layout of the engine, against the oracle (`pytest -m gpu`).

The fixed cases of test_gpu_parity.py are chosen by hand; these are not: row counts from 1 to a few hundred thousand,
column counts from 1 to millions, row lengths constant / Poisson / power-law with empty rows, columns uniform / banded /
clustered, every CSR kernel (row-parallel, one lane per row, LDS window where it fits, panel in each of its layouts,
chunk sizes, pipelines and wavefront syncs, two-phase with several panel widths), ELL handles from clean stencils to
noise, COO in file order with duplicates.  Gate: SURVEY 8d's two tolerances for kernels that reorder sums, equality bit
for bit with the fma oracle for the kernels that keep the reference's order (ELL, one lane per row)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


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
    rng = np.random.default_rng(1000 + seed)
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
                         ("panel_width", width), ("panel_pace_ns", 0)):
                A.set_param(k, v)
            A.set_kernel(capi.CSR_PANEL)
            run(A, f"panel layout={layout}->{A.get_param('panel_layout')} unroll={unroll} pipe={pipe} sync={sync} rows={rows} width={width}")
        if nnz + 16 * ((ncol + 6999) // 7000) * 256 < 2**31:
            for cols, unroll in ((20_000, 6), (7_000, 4)):
                A.set_param("twophase_panel_cols", cols)
                A.set_param("twophase_unroll", unroll)
                A.set_kernel(capi.CSR_TWOPHASE)
                run(A, f"two-phase cols={cols} unroll={unroll}")


@pytest.mark.parametrize("seed", range(24))
def test_random_ell_handles_from_stencils_to_noise(ctx, orc, pkg, seed):
    """column-major ELL with a random mix of diagonal slots, padding and arbitrary columns: the diagonal-slot kernel (when
    the analysis takes it) and the plain ones must equal the fma oracle exactly"""
    rng = np.random.default_rng(2000 + seed)
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
    for flags in (0, 8):
        A.set_flags(flags)
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        assert np.array_equal(dy.download(), ref), (seed, nrow, ncol, k, flags, A.get_param("ell_diagonal_slots"))


@pytest.mark.parametrize("seed", range(16))
def test_random_coo_in_file_order_with_duplicates(ctx, orc, pkg, seed):
    capi = pkg.capi
    rng = np.random.default_rng(3000 + seed)
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
    C = ctx.coo_to_csr(A)
    assert C.info.nnz == nnz
    got = C.download()
    assert np.array_equal(got[0], rp) and np.array_equal(got[1], cc) and np.array_equal(got[2], cv)
