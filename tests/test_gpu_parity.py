"""GPU parity tests proper: the HIP engine, called through the C ABI, against the oracle and the golden
vectors produced by the real reference.  Run on the MI355X box with `pytest -m gpu`.

Tolerance (BASELINE.json north_star): after 1 call from y = 0 and after 50 accumulating calls
    max|y - y_ref| / max|y_ref| <= 1e-10   and   max_i |y_i - y_ref,i| / (|A||x|)_i <= 1e-10.
Kernels that add in the reference's order (ELL, scalar CSR, DIA) are additionally bit-identical to the
oracle's fused-multiply-add flavour; integer work (conversions, generators, sharding) is bit-exact.
"""
import numpy as np
import pytest

import cases
import oracle_lib as ol
from conftest import golden, perf_expect

pytestmark = pytest.mark.gpu
NUM_TEST = 50  # main.cpp:16


def _csr_of(orc, c):
    return ol.coo_to_csr(orc, c["nrow"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]))


def _scale(orc, c):
    rp, cc, cv = _csr_of(orc, c)
    s = np.zeros(c["nrow"])
    ol.csr_abs_row_sums(orc, rp, cc, cv, ol.f64(c["x"]), s)
    return s


def _apply_n(ctx, A, x, nrow, reps):
    dx = ctx.vector_from(x)
    dy = ctx.vector(nrow)
    dy.fill(0.0)
    out = []
    for r in range(1, reps + 1):
        ctx.apply(A, dx, dy)
        if r in (1, reps):
            ctx.sync()
            out.append(dy.download())
    return out[0], out[-1]


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_csr_matches_reference_golden(ctx, orc, pkg, make):
    c = make()
    g = golden(c["name"])
    rp, cc, cv = _csr_of(orc, c)
    scale = _scale(orc, c)
    capi = pkg.capi
    for kernel, lanes in ((capi.CSR_AUTO, 0), (capi.CSR_VECTOR, 1), (capi.CSR_VECTOR, 2), (capi.CSR_VECTOR, 4),
                          (capi.CSR_VECTOR, 8), (capi.CSR_VECTOR, 16), (capi.CSR_VECTOR, 32), (capi.CSR_VECTOR, 64),
                          (capi.CSR_SCALAR, 0), (capi.CSR_TWOPHASE, 0), (capi.CSR_SEGSCAN, 0), (capi.CSR_SPLIT, 0), (capi.CSR_ELL, 0)):
        for flags in (0, capi.FLAG_DPP_REDUCE, capi.FLAG_XCD_REMAP):
            if kernel != capi.CSR_VECTOR and flags:
                continue
            A = ctx.csr(c["nrow"], c["ncol"], rp, cc, cv)
            try:
                A.set_kernel(kernel, lanes)
            except capi.SpmvError as e:  # the ELL copy of one 4096-entry row among short ones, or of a matrix with an empty row: refused, not built
                assert kernel == capi.CSR_ELL and ("out of proportion" in str(e) or "empty row" in str(e)), e
                continue
            A.set_flags(flags)
            y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
            what = f"{c['name']} csr kernel={kernel} lanes={lanes} flags={flags}"
            ol.assert_parity(y1, g["y1_csr"], scale, what + " 1 call")
            ol.assert_parity(y50, g["y50_csr"], scale, what + " 50 calls", reps=NUM_TEST)


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_csr_scalar_kernel_is_bitwise_oracle_fma(ctx, orc, pkg, make):
    c = make()
    rp, cc, cv = _csr_of(orc, c)
    A = ctx.csr(c["nrow"], c["ncol"], rp, cc, cv)
    A.set_kernel(pkg.capi.CSR_SCALAR)
    y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
    ref = np.zeros(c["nrow"])
    x = ol.f64(c["x"])
    ol.csr_spmv(orc, rp, cc, cv, x, ref, fma=True)
    assert np.array_equal(y1, ref)
    for _ in range(NUM_TEST - 1):
        ol.csr_spmv(orc, rp, cc, cv, x, ref, fma=True)
    assert np.array_equal(y50, ref)


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_ell_matches_reference_and_is_bitwise_oracle_fma(ctx, orc, make):
    c = make()
    g = golden(c["name"])
    k, ec, ev = ol.coo_to_ell(orc, c["nrow"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]))
    assert k == int(g["ell_k"])
    scale = _scale(orc, c)
    A = ctx.ell(c["nrow"], c["ncol"], k, len(c["val"]), ec, ev)
    y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
    ol.assert_parity(y1, g["y1_ell"], scale, c["name"] + " ell 1 call")
    ol.assert_parity(y50, g["y50_ell"], scale, c["name"] + " ell 50 calls", reps=NUM_TEST)
    ref = np.zeros(c["nrow"])
    ol.ell_spmv(orc, c["nrow"], k, ec, ev, ol.f64(c["x"]), ref, fma=True)
    # (AUTO may run a handle of few long rows from its row-grouped copy - c1: 10000 rows - within the gate above; the format's
    # own kernels add in the reference's per-row order)
    for lanes in (1, 2):
        A.set_kernel(1, lanes)
        y1, _ = _apply_n(ctx, A, c["x"], c["nrow"], 1)
        assert np.array_equal(y1, ref), "ELL kernel adds in the reference's per-row order: must equal the fma oracle exactly"


def test_ell_odd_row_count_uses_one_row_kernel(ctx, orc):
    """nrow odd -> the 16-byte two-rows-per-lane kernel is not applicable; same answers"""
    rng = np.random.RandomState(3)
    nrow, ncol, k = 1001, 777, 5
    col = rng.randint(0, ncol, size=nrow * k).astype(np.int32)
    val = rng.uniform(-1, 1, size=nrow * k)
    x = rng.uniform(0, 1, size=ncol)
    A = ctx.ell(nrow, ncol, k, nrow * k, col, val)
    y1, _ = _apply_n(ctx, A, x, nrow, 1)
    ref = np.zeros(nrow)
    ol.ell_spmv(orc, nrow, k, col, val, x, ref, fma=True)
    assert np.array_equal(y1, ref)


@pytest.mark.parametrize("shape", ["stencil", "wide_band", "circulant", "mostly_irregular"])
def test_ell_slots_that_are_diagonals_need_no_column_stream(ctx, orc, pkg, shape):
    """ELL handles of stencil / band matrices: slot s of (nearly) every row holds column i + off[s]; the product then
    reads no column index for those rows and takes x through an LDS window.  Rows that do not conform (boundary rows
    padded with (0, 0.0), wrap-around rows, a few rows with arbitrary columns) read their columns as before.  Same
    products, same order: equal to the fma oracle bit for bit, with the detection on and switched off."""
    rng = np.random.RandomState(11)
    if shape == "stencil":  # 5-point-like: offsets -70, -1, 0, 1, 70; boundary entries are padding
        nrow = ncol = 6000
        offs = np.array([-70, -1, 0, 1, 70])
    elif shape == "wide_band":  # offsets far apart: one stretch of x in LDS per cluster of nearby offsets (four here)
        nrow = ncol = 30_000
        offs = np.array([-9000, -3, 0, 3, 9000, 12_000])
    elif shape == "circulant":  # wraps around like C3; rectangular: more columns than rows
        nrow, ncol = 4096, 5000
        offs = np.arange(-6, 7)
    else:  # most rows arbitrary: the detection must decline
        nrow = ncol = 5000
        offs = np.array([-2, 0, 2, 5])
    k = len(offs)
    rows = np.arange(nrow)
    col = rows[None, :] + offs[:, None]  # [slot, row]: column-major ELL (include/matrix.h:70)
    val = rng.uniform(-1, 1, size=(k, nrow))
    if shape == "circulant":
        col = col % ncol
    else:
        out = (col < 0) | (col >= ncol)
        col[out], val[out] = 0, 0.0  # padding as the reference's constructor writes it (src/matrix.cpp:473-474)
    odd = rng.choice(nrow, size=(nrow * 9 // 10 if shape == "mostly_irregular" else 25), replace=False)
    if shape != "mostly_irregular":
        odd = odd[odd != nrow // 2]  # the detection reads the offsets off the middle row
    col[:, odd] = rng.randint(0, ncol, size=(k, len(odd)))  # rows with arbitrary columns
    col, val = col.astype(np.int32).ravel(), val.ravel()
    x = rng.uniform(0, 1, size=ncol)
    ref = np.zeros(nrow)
    ol.ell_spmv(orc, nrow, k, col, val, x, ref, fma=True)
    A = ctx.ell(nrow, ncol, k, nrow * k, col, val)
    assert A.get_param("ell_diagonal_slots") == (0 if shape == "mostly_irregular" else 1)
    for flags in (0, 8):  # 8 = SPMV_FLAG_ELL_READ_COLUMNS
        A.set_flags(flags)
        y1, y50 = _apply_n(ctx, A, x, nrow, NUM_TEST)
        assert np.array_equal(y1, ref), (shape, flags)
    # the values once more in tiles of 512 rows (what a workgroup reads becomes one contiguous stretch; opt-in): same loads in
    # the same order, so the same bits - rows that do not conform read their columns from the column-major array as before
    A.set_flags(0)
    assert A.get_param("ell_tiled_values") == 0
    held = A.get_param("device_bytes")
    A.set_param("ell_tiled_values", 1)
    if shape == "mostly_irregular":
        assert A.get_param("ell_tiled_values") == 0 and A.get_param("device_bytes") == held  # (no diagonals: nothing to tile for)
    else:
        assert A.get_param("ell_tiled_values") == 1 and A.get_param("device_bytes") == held + 8 * k * 512 * -(-nrow // 512)
        y1, y50 = _apply_n(ctx, A, x, nrow, NUM_TEST)
        assert np.array_equal(y1, ref), (shape, "tiled")
        got = A.download()  # (the handle's own arrays are what they were)
        assert np.array_equal(got[-1], val) and np.array_equal(got[-2], col)
    # x with NaN where only padding looks (x[0] times 0.0 poisons the padded rows in the reference as well: keep it)
    if shape == "stencil":
        xn = x.copy()
        xn[0] = np.inf
        refn = np.zeros(nrow)
        ol.ell_spmv(orc, nrow, k, col, val, xn, refn, fma=True)
        A.set_flags(0)
        yn, _ = _apply_n(ctx, A, xn, nrow, 1)
        assert np.array_equal(np.isnan(yn), np.isnan(refn)) and np.array_equal(yn[~np.isnan(refn)], refn[~np.isnan(refn)])
    A.set_param("ell_tiled_values", 0)
    assert A.get_param("ell_tiled_values") == 0 and A.get_param("device_bytes") == held
    y1, _ = _apply_n(ctx, A, x, nrow, 1)
    assert np.array_equal(y1, ref), (shape, "tiles dropped")
    # Round 6: the values once more in DIA ORDER (row-major) under the DIA kernel - a workgroup streams one contiguous stretch and
    # x goes through an LDS window where the offsets lie within 1792 of each other; rows in which any slot is not its diagonal
    # (padding, wrap-around, arbitrary columns) are skipped there and done by a side kernel over the column-major arrays.  The
    # same products in the same order: the fma oracle bit for bit.  8 bytes per slot; a candidate of AUTO's trial from 1M slots on
    capi = pkg.capi
    if shape == "mostly_irregular":
        with pytest.raises(capi.SpmvError, match="diagonals"):
            A.set_param("ell_dia_order", 1)
        assert A.get_param("ell_dia_order") == 0 and A.get_param("device_bytes") == held
        return
    A.set_param("ell_dia_order", 1)
    nc = A.get_param("ell_non_conforming_rows")
    conforming = np.all(col.reshape(k, nrow) == (np.arange(nrow)[None, :] + offs[:, None]), axis=0)
    assert A.get_param("ell_dia_order") == 1 and A.get_param("ell_variant") == 3 and nc == int((~conforming).sum()) >= len(odd)
    assert A.get_param("device_bytes") == held + 8 * (k + k % 2) * nrow + 8 * -(-nrow // 64) + 4 * max(nc, 1)  # (rows of the copy padded to an even stride)
    y1, y50 = _apply_n(ctx, A, x, nrow, NUM_TEST)
    assert np.array_equal(y1, ref), (shape, "DIA order")
    ref50 = np.zeros(nrow)
    for _ in range(NUM_TEST):
        ol.ell_spmv(orc, nrow, k, col, val, x, ref50, fma=True)
    assert np.array_equal(y50, ref50), (shape, "DIA order, 50 calls")
    if shape == "stencil":  # the padding's 0.0 * x[0] with x[0] = inf: padded rows are non-conforming rows, done in slot order
        yn, _ = _apply_n(ctx, A, xn, nrow, 1)
        assert np.array_equal(np.isnan(yn), np.isnan(refn)) and np.array_equal(yn[~np.isnan(refn)], refn[~np.isnan(refn)])
    got = A.download()
    assert np.array_equal(got[-1], val) and np.array_equal(got[-2], col)
    # it travels in a plan: another handle of the same matrix built from it runs the same variant and gives the same bits
    plan = A.get_plan()
    B = ctx.ell(nrow, ncol, k, nrow * k, col, val)
    assert B.get_param("ell_dia_order") == 0  # (180K slots at most: no candidate of the trial)
    B.set_plan(plan)
    assert B.get_param("ell_dia_order") == 1 and B.get_plan() == plan
    yb, _ = _apply_n(ctx, B, x, nrow, 1)
    assert np.array_equal(yb, ref), (shape, "DIA order from a plan")
    A.set_param("ell_dia_order", 0)
    assert A.get_param("ell_dia_order") == 0 and A.get_param("ell_variant") == 0 and A.get_param("device_bytes") == held
    y1, _ = _apply_n(ctx, A, x, nrow, 1)
    assert np.array_equal(y1, ref), (shape, "DIA-order copy dropped")


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_coo_matches_reference_golden(ctx, orc, pkg, make):
    c = make()
    g = golden(c["name"])
    scale = _scale(orc, c)
    A = ctx.coo(c["nrow"], c["ncol"], c["row"], c["col"], c["val"])
    sorted_in = bool(np.all(np.diff(c["row"].astype(np.int64)) >= 0))
    assert bool(A.info.sorted_rows) == sorted_in
    # AUTO: below 64K entries nothing is timed and the segmented scan runs; above, the scan and the row-grouped copy are
    # timed and the faster one stays (C1 has 160K entries)
    import os

    trials = os.environ.get("SPMV_PANEL_TRIAL", "1")[:1] != "0"  # (tools/env_sweeps.sh runs the suite with the model alone too)
    if len(c["val"]) < 65536 or not trials:
        assert A.info.kernel == pkg.capi.CSR_VECTOR and A.get_param("select_candidates") == 0
    else:
        assert A.info.kernel in (pkg.capi.CSR_VECTOR, pkg.capi.CSR_PANEL) and A.get_param("select_candidates") == 2
        y1, _ = _apply_n(ctx, A, c["x"], c["nrow"], 1)
        ol.assert_parity(y1, g["y1_coo"], scale, f"{c['name']} coo AUTO (kernel {A.info.kernel}, copy runs {A.get_param('rowgrouped_kernel')})")
    for kernel in (pkg.capi.CSR_VECTOR, pkg.capi.CSR_PANEL):  # PANEL: grouped by row on the device, panel layout
        A.set_kernel(kernel)
        assert A.info.kernel == kernel
        y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
        ol.assert_parity(y1, g["y1_coo"], scale, f"{c['name']} coo kernel={kernel} 1 call")
        ol.assert_parity(y50, g["y50_coo"], scale, f"{c['name']} coo kernel={kernel} 50 calls", reps=NUM_TEST)


def test_coo_ragged_chunk_boundaries(ctx, orc):
    """runs that straddle wavefront chunks (512 entries) and workgroup chunks (2048), sorted and shuffled"""
    rng = np.random.RandomState(11)
    nrow, ncol = 4000, 3000
    lens = rng.choice([0, 1, 2, 3, 63, 64, 65, 511, 512, 513, 700, 2047, 2049], size=nrow, p=[.3, .2, .1, .1, .05, .05, .05, .03, .03, .03, .03, .02, .01])
    row = np.repeat(np.arange(nrow, dtype=np.int32), lens)
    col = rng.randint(0, ncol, size=row.size).astype(np.int32)
    val = rng.uniform(-1, 1, size=row.size)
    x = rng.uniform(0, 1, size=ncol)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    scale = np.zeros(nrow)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    for shuffle in (False, True):
        if shuffle:
            p = rng.permutation(row.size)
            row, col, val = row[p], col[p], val[p]
        ref = np.zeros(nrow)
        ol.coo_spmv(orc, row, col, val, x, ref)
        A = ctx.coo(nrow, ncol, row, col, val)
        assert bool(A.info.sorted_rows) == (not shuffle)
        y1, _ = _apply_n(ctx, A, x, nrow, 1)
        ol.assert_parity(y1, ref, scale, f"coo ragged shuffle={shuffle}")


@pytest.mark.parametrize("ncol,skew", [(3000, False), (1, False), (70_000, True), (5_000_000, False)])
def test_coo_segmented_scan_over_column_bins(ctx, orc, pkg, ncol, skew):
    """The segmented scan over the copy of the entries in column bins (one run of bins per XCD): 1, 3 and 8 bins
    per XCD, sorted and shuffled input, columns uniform or with half of the entries in one column (the bins are entry
    quantiles: some stay empty), against the oracle; dropping the copy gives the bytes back and the scan in place agrees."""
    capi = pkg.capi
    rng = np.random.RandomState(ncol % 1000 + 5)
    nrow = 4000
    lens = rng.choice([0, 1, 2, 3, 63, 64, 65, 511, 512, 513, 700, 2047, 2049], size=nrow, p=[.3, .2, .1, .1, .05, .05, .05, .03, .03, .03, .03, .02, .01])
    row = np.repeat(np.arange(nrow, dtype=np.int32), lens)
    col = rng.randint(0, ncol, size=row.size).astype(np.int32)
    if skew:
        col[rng.rand(row.size) < 0.5] = 12345
    val = rng.uniform(-1, 1, size=row.size)
    x = rng.uniform(0, 1, size=ncol)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    ref, scale = np.zeros(nrow), np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    for shuffle in (False, True):
        if shuffle:
            p = rng.permutation(row.size)
            row, col, val = row[p], col[p], val[p]
        A = ctx.coo(nrow, ncol, row, col, val)
        A.set_kernel(capi.CSR_VECTOR)
        assert A.get_param("coo_column_bins") == 0  # (too small for the copy to be made unasked)
        held = A.get_param("device_bytes")
        for per_xcd in (1, 3, 8):
            A.set_param("coo_column_bins", per_xcd)
            assert A.get_param("coo_column_bins") == 8 * per_xcd
            padded = A.get_param("coo_bins_padded")
            assert padded % 2048 == 0 and row.size <= padded <= row.size + 8 * per_xcd * 2048
            assert A.get_param("device_bytes") == held + 16 * padded
            y1, y3 = _apply_n(ctx, A, x, nrow, 3)
            ol.assert_parity(y1, ref, scale, f"coo over {8 * per_xcd} column bins, shuffle={shuffle}")
            ol.assert_parity(y3, 3 * ref, 3 * scale, f"coo over {8 * per_xcd} column bins, three products, shuffle={shuffle}")
        A.set_param("coo_column_bins", 0)
        assert A.get_param("coo_column_bins") == 0 and A.get_param("device_bytes") == held
        y1, _ = _apply_n(ctx, A, x, nrow, 1)
        ol.assert_parity(y1, ref, scale, f"coo in place after the copy was dropped, shuffle={shuffle}")


def test_empty_and_degenerate_inputs(ctx):
    x = np.ones(5)
    for A in (ctx.csr(0, 5, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0)),
              ctx.coo(0, 5, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)),
              ctx.ell(0, 5, 0, 0, np.zeros(0, np.int32), np.zeros(0))):
        dy = ctx.vector(0)
        ctx.apply(A, ctx.vector_from(x), dy)
        ctx.sync()
    # rows but no entries: y must stay what it was
    A = ctx.csr(7, 5, np.zeros(8, np.int32), np.zeros(0, np.int32), np.zeros(0))
    dy = ctx.vector_from(np.arange(7.0))
    ctx.apply(A, ctx.vector_from(x), dy)
    ctx.sync()
    assert np.array_equal(dy.download(), np.arange(7.0))
    A = ctx.coo(7, 5, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    ctx.apply(A, ctx.vector_from(x), dy)
    ctx.sync()
    assert np.array_equal(dy.download(), np.arange(7.0))


def test_apply_rejects_shape_mismatch(ctx, pkg):
    A = ctx.csr(2, 3, np.array([0, 1, 2], np.int32), np.array([0, 2], np.int32), np.array([1.0, 2.0]))
    with pytest.raises(pkg.capi.SpmvError):
        ctx.apply(A, ctx.vector(2), ctx.vector(2))  # x must have ncol = 3 entries
    with pytest.raises(pkg.capi.SpmvError):
        ctx.apply(A, ctx.vector(3), ctx.vector(3))  # y must have nrow = 2 entries


# ---------------------------------------------------------------------------------- conversions (bit-exact)
@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_coo_to_csr_and_ell_equal_reference_arrays(ctx, orc, make):
    c = make()
    g = golden(c["name"])
    coo = ctx.coo(c["nrow"], c["ncol"], c["row"], c["col"], c["val"])
    csr = ctx.coo_to_csr(coo)
    rp, cc, cv = csr.download()
    ell = ctx.coo_to_ell(coo)
    _, ec, ev = ell.download()
    assert ell.info.ell_k == int(g["ell_k"])
    if c["name"] == "c1":
        assert cases.digest(rp, cc, cv) == str(g["sha_csr"])
        assert cases.digest(ec, ev) == str(g["sha_ell"])
    else:
        assert np.array_equal(rp, g["csr_row_ptr"]) and np.array_equal(cc, g["csr_col"]) and np.array_equal(cv, g["csr_val"])
        assert np.array_equal(ec, g["ell_col"]) and np.array_equal(ev, g["ell_val"])
    ell2 = ctx.csr_to_ell(csr)
    _, ec2, ev2 = ell2.download()
    assert np.array_equal(ec, ec2) and np.array_equal(ev, ev2)


def test_coo_to_csr_large_unsorted(ctx, orc):
    rng = np.random.RandomState(5)
    nrow, ncol, nnz = 50_000, 40_000, 700_000
    row = rng.randint(0, nrow, size=nnz).astype(np.int32)
    col = rng.randint(0, ncol, size=nnz).astype(np.int32)
    val = rng.uniform(-1, 1, size=nnz)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    got = ctx.coo_to_csr(ctx.coo(nrow, ncol, row, col, val)).download()
    assert np.array_equal(got[0], rp) and np.array_equal(got[1], cc) and np.array_equal(got[2], cv)


def test_coo_to_csr_unsorted_hub_row(ctx, orc):
    """one shuffled row of 300 000 entries among ordinary ones, unsorted input: the conversion must keep the COO order
    inside the hub row (src/matrix.cpp:140-144) and finish promptly (rank-by-scanning would be ~10^11 steps there)"""
    import time

    rng = np.random.RandomState(9)
    nrow, ncol, hub = 20_000, 400_000, 300_000
    row = np.concatenate([np.full(hub, 1234, np.int32), rng.randint(0, nrow, size=200_000).astype(np.int32)])
    col = rng.randint(0, ncol, size=row.size).astype(np.int32)
    val = rng.uniform(-1, 1, size=row.size)
    perm = rng.permutation(row.size)
    row, col, val = row[perm], col[perm], val[perm]
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    t = time.perf_counter()
    got = ctx.coo_to_csr(ctx.coo(nrow, ncol, row, col, val)).download()
    assert time.perf_counter() - t < 20.0
    assert np.array_equal(got[0], rp) and np.array_equal(got[1], cc) and np.array_equal(got[2], cv)


# ---------------------------------------------------------------------------------- CSC / DIA ("next" rows)
@pytest.mark.parametrize("make", cases.SMALL_CASES, ids=lambda f: f.__name__)
def test_csc_matches_reference_golden(ctx, orc, make):
    c = make()
    g = golden(c["name"])
    A = ctx.csc(c["nrow"], c["ncol"], g["csc_col_ptr"], g["csc_row"], g["csc_val"])
    y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
    scale = _scale(orc, c)
    ol.assert_parity(y1, g["y1_csc"], scale, c["name"] + " csc 1 call")
    ol.assert_parity(y50, g["y50_csc"], scale, c["name"] + " csc 50 calls", reps=NUM_TEST)


def test_dia_matches_reference_golden(ctx, orc):
    c = cases.tri8()
    g = golden("tri8")
    A = ctx.dia(c["nrow"], c["ncol"], g["dia_offsets"], g["dia_val"])
    y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
    scale = _scale(orc, c)
    ol.assert_parity(y1, g["y1_dia"], scale, "dia 1 call")
    ol.assert_parity(y50, g["y50_dia"], scale, "dia 50 calls", reps=NUM_TEST)
    ref = np.zeros(c["nrow"])
    ol.dia_spmv(orc, c["nrow"], ol.i32(g["dia_offsets"]), ol.f64(g["dia_val"]), ol.f64(c["x"]), ref, fma=True)
    assert np.array_equal(y1, ref)


@pytest.mark.parametrize("tall", [True, False])
def test_dia_rectangular_bounds_columns_by_min_nrow_ncol(ctx, orc, tall):
    """rect64x48 and its transpose through DIA.  The reference bounds columns by nrow (src/mat_vec.cpp:140): for a
    tall matrix it reads x[ncol..nrow) (times stored zeros), for a wide one it drops columns >= nrow.  The oracle
    gets x padded with zeros to max(nrow, ncol), which makes its overread defined; the kernel must give the same
    sums without touching anything past x's ncol entries (x is the last allocation made before the product and
    non-finite padding would show)."""
    c = cases.rect64x48()
    nrow, ncol, row, col = (c["nrow"], c["ncol"], c["row"], c["col"]) if tall else (c["ncol"], c["nrow"], c["col"], c["row"])
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, c["val"])
    off, dv = ol.csr_to_dia(orc, nrow, ncol, rp, cc, cv)
    x = np.random.default_rng(12).uniform(0.5, 1.5, ncol)
    xpad = np.zeros(max(nrow, ncol))
    xpad[:ncol] = x
    ref = np.zeros(nrow)
    ol.dia_spmv(orc, nrow, ol.i32(off), ol.f64(dv), xpad, ref, fma=True)
    A = ctx.dia(nrow, ncol, off, dv)
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    dy.fill(0.0)
    ctx.apply(A, dx, dy)
    ctx.sync()
    assert np.array_equal(dy.download(), ref), ("tall" if tall else "wide")


def test_csr_handle_can_give_up_its_arrays_once_the_panel_layout_is_built(ctx, pkg):
    """panel_keep_csr = 0: col_ind / values of the CSR copy are released (the panel layout holds the same entries),
    memory drops to ~1x the matrix, the product is unchanged, and what needs the arrays is refused with a message"""
    capi = pkg.capi
    n, k = 1_000_000, 16
    A = ctx.gen_csr_uniform(0, n, n, k, seed=31)
    assert A.info.kernel == capi.CSR_PANEL and A.get_param("panel_layout") == 4
    x, y0, y1 = ctx.gen_vector(n, seed=31), ctx.vector(n), ctx.vector(n)
    y0.fill(0.0)
    y1.fill(0.0)
    ctx.apply(A, x, y0)
    before = A.get_param("device_bytes")
    free0, _ = ctx.mem_info()
    A.set_param("panel_keep_csr", 0)
    after = A.get_param("device_bytes")
    free1, _ = ctx.mem_info()
    nnz = n * k
    assert before - after == 12 * nnz and after <= 1.1 * (12 * nnz + 4 * (n + 1)), (before, after)
    assert free1 - free0 >= 0.9 * 12 * nnz  # the driver really got the memory back
    assert A.get_param("panel_keep_csr") == 0
    ctx.apply(A, x, y1)
    ctx.sync()
    assert np.allclose(y0.download(), y1.download(), rtol=0, atol=1e-10 * k)
    Err = capi.SpmvError
    with pytest.raises(Err, match="gave up"):
        A.download()
    with pytest.raises(Err, match="gave up"):
        A.set_kernel(capi.CSR_VECTOR)
    with pytest.raises(Err, match="gave up"):
        ctx.csr_to_ell(A)
    with pytest.raises(Err, match="gave up"):
        A.validate()  # refused on the host: the check kernels would read the released index array
    A.set_param("panel_rows", 5000)
    with pytest.raises(Err, match="gave up"):
        A.set_kernel(capi.CSR_PANEL)  # a re-build with other parameters needs the CSR arrays
    A.set_param("panel_rows", 0)
    A.set_kernel(capi.CSR_PANEL)  # same parameters: nothing to re-build
    ctx.apply(A, x, y1)
    ctx.sync()


def test_csr_twophase_kernel_on_wide_and_ragged_matrices(ctx, orc, pkg):
    """expand (x panels in LDS) + reduce (row groups in LDS): several panels and groups, empty rows, a hub row, a last
    panel that is not full, accumulation over two calls, and the solver's fused overwrite + dot; against the
    row-parallel kernel and the oracle's row sums as the scale"""
    capi = pkg.capi
    rng = np.random.RandomState(17)
    nrow, ncol = 45_000, 130_001  # 3 groups, 7 panels (the last one 10001 columns wide)
    lens = rng.randint(0, 40, size=nrow)
    lens[::97] = 0
    lens[12345] = 6000
    rp = np.zeros(nrow + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    cc = rng.randint(0, ncol, size=nnz).astype(np.int32)
    cv = rng.uniform(-1, 1, size=nnz)
    x = rng.uniform(0, 1, size=ncol)
    ref = np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    scale = np.zeros(nrow)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.csr(nrow, ncol, rp, cc, cv)
    A.set_kernel(capi.CSR_TWOPHASE)
    assert A.info.kernel == capi.CSR_TWOPHASE
    dx, dy = ctx.vector_from(x), ctx.vector(nrow)
    dy.fill(0.0)
    ctx.apply(A, dx, dy)
    ctx.sync()
    ol.assert_parity(dy.download(), ref, scale, "twophase 1 call")
    ctx.apply(A, dx, dy)
    ctx.sync()
    ol.assert_parity(dy.download(), 2 * ref, scale, "twophase 2 calls", reps=2)
    w = rng.uniform(-1, 1, size=nrow)
    dw = ctx.vector_from(w)
    d = ctx.apply_dot(A, dx, dy, dw, overwrite=True)
    got = dy.download()
    ol.assert_parity(got, ref, scale, "twophase overwrite")
    assert abs(d - float(w @ got)) <= 1e-9 * float(np.abs(w) @ np.abs(got))
    # x that is only 8-byte aligned (a view one entry into a vector): the expand phase then loads its panels of x with
    # 8-byte instead of 16-byte loads
    big = ctx.vector(ncol + 1)
    big.upload(np.concatenate(([123.0], x)))
    dx_odd = ctx.wrap_vector(big.device_ptr + 8, ncol)
    dy.fill(0.0)
    ctx.apply(A, dx_odd, dy)
    ctx.sync()
    ol.assert_parity(dy.download(), ref, scale, "twophase, x not 16-byte aligned")
    # narrower panels (an odd request is rounded down to even: x moves in pairs) and every run padding
    # (pairs / 64-byte pieces / 128-byte lines: boundaries at any pair of a source line, at half lines, at lines only)
    import os

    for cols, unroll, pad in ((10_000, 3, 8), (7_001, 3, 8), (20_000, 3, 2), (333, 3, 2), (20_000, 3, 16), (4_000, 3, 16)):
        os.environ["SPMV_TP_PAD"] = str(pad)
        A.set_param("twophase_panel_cols", cols)
        A.set_kernel(capi.CSR_TWOPHASE)
        os.environ.pop("SPMV_TP_PAD")
        assert A.get_param("twophase_panel_cols") == cols - cols % 2 and A.get_param("twophase_padded") % pad == 0
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, f"twophase panel {cols} unroll {unroll}")
    # the automatic choice: a shard far wider than tall takes it when its runs (panel x row group) are long enough to
    # pad to whole lines; a sparser one and a square matrix of the same size do not
    # (the MODEL's choice: between 8M and 64M entries a handle it sends to the two phases also times the panel layout, select.hip)
    os.environ["SPMV_PANEL_TRIAL"] = "0"
    wide = ctx.gen_csr_uniform(0, 2_500_000, 40_000_000, 16, seed=3)
    os.environ.pop("SPMV_PANEL_TRIAL")
    assert wide.info.kernel == capi.CSR_TWOPHASE
    assert wide.info.nnz <= wide.get_param("twophase_padded") <= 1.25 * wide.info.nnz
    xw, yw, yv = ctx.gen_vector(40_000_000, seed=3), ctx.vector(2_500_000), ctx.vector(2_500_000)
    yw.fill(0.0)
    yv.fill(0.0)
    ctx.apply(wide, xw, yw)
    wide.set_kernel(capi.CSR_VECTOR)
    ctx.apply(wide, xw, yv)
    ctx.sync()
    assert np.max(np.abs(yw.download() - yv.download())) <= ol.REL_TOL * 16
    del wide
    # (handles below 8M entries also TIME their candidates: the model's own answer is what SPMV_PANEL_TRIAL=0 leaves)
    os.environ["SPMV_PANEL_TRIAL"] = "0"
    thin = ctx.gen_csr_uniform(0, 600_000, 40_000_000, 8, seed=3)
    os.environ.pop("SPMV_PANEL_TRIAL")
    assert thin.info.kernel == capi.CSR_PANEL and thin.get_param("select_candidates") == 0
    del thin
    # few entries per row would pass the sweep model, but a band matrix only sweeps its band: the panel kernel stays
    band = ctx.gen_csr_uniform(0, 3_000_000, 3_000_000, 4, band=4096, seed=3)
    assert band.info.kernel == capi.CSR_PANEL
    del band
    os.environ["SPMV_PANEL_TRIAL"] = "0"
    sparse = ctx.gen_csr_uniform(0, 3_000_000, 3_000_000, 4, seed=3)  # the same shape with uniform columns takes the two phases
    os.environ.pop("SPMV_PANEL_TRIAL")
    assert sparse.info.kernel == capi.CSR_TWOPHASE
    del sparse
    # ... and with the trials on, the two phases and the panel layout are both timed there and the faster one stays
    sparse = ctx.gen_csr_uniform(0, 3_000_000, 3_000_000, 4, seed=3)
    t2, tp = sparse.get_param("select_us_twophase"), sparse.get_param("select_us_panel")
    # (exactly 4 entries in every row, but scattered columns: the ELL copy of csr_ell_copy_worth is no candidate)
    assert sparse.get_param("select_candidates") == 2 and t2 > 0 and tp > 0 and sparse.get_param("select_us_ell") == 0
    assert sparse.info.kernel in (capi.CSR_TWOPHASE, capi.CSR_PANEL)
    perf_expect(sparse.info.kernel == (capi.CSR_TWOPHASE if t2 <= tp else capi.CSR_PANEL) or abs(t2 - tp) <= 0.03 * tp + 1, f"two phases {t2} us, panel {tp} us, kept {sparse.info.kernel}")
    del sparse
    os.environ["SPMV_PANEL_TRIAL"] = "0"
    square = ctx.gen_csr_uniform(0, 600_000, 600_000, 8, seed=3)
    os.environ.pop("SPMV_PANEL_TRIAL")
    assert square.info.kernel == capi.CSR_PANEL


def test_native_exchange_allgather_and_vec_copy(pkg):
    """spmv_comm_* with three contexts of this process (on a one-GPU box they share the device: the copies are
    device-to-device; with a GPU each they are RCCL broadcasts or peer copies): ragged slices, repeated calls with the
    slices rewritten in between, and spmv_vec_copy between contexts"""
    capi = pkg.capi
    ctxs = [capi.Context(0) for _ in range(3)]
    comm = capi.Comm(ctxs)
    assert comm.backend in ("peer-copy", "rccl")
    n = 1_000_003
    offsets = np.array([0, 333_334, 333_334 + 400_000, n], dtype=np.int64)  # ragged, like the reference's last shard
    vecs = [c.vector(n) for c in ctxs]
    rng = np.random.default_rng(3)
    for rep in range(3):
        full = rng.uniform(-1, 1, n)
        for i, v in enumerate(vecs):
            v.fill(float("nan"))  # whatever is not the participant's own slice must come from the others
            v.upload(full[offsets[i]:offsets[i + 1]], offset=int(offsets[i]))
        comm.allgather(vecs, offsets)
        for c in ctxs:
            c.sync()
        for v in vecs:
            assert np.array_equal(v.download(), full), rep
    # one participant: nothing to move, nothing to break
    solo = capi.Comm(ctxs[:1])
    solo.allgather(vecs[:1], np.array([0, n], dtype=np.int64))
    # copy between contexts, ordered behind the source's queued work
    dst = ctxs[2].vector(50)
    dst.fill(0.0)
    vecs[0].fill(7.0)
    dst.copy_from(vecs[0], 20, dst_offset=10, src_offset=12345)
    ctxs[2].sync()
    got = dst.download()
    assert np.array_equal(got[10:30], np.full(20, 7.0)) and not got[:10].any() and not got[30:].any()
    with pytest.raises(capi.SpmvError):
        dst.copy_from(vecs[0], 60)


def test_rccl_transport_runs_with_one_participant_on_one_gpu(pkg, monkeypatch):
    """SPMV_COMM=rccl forces the RCCL transport of spmv_comm_* (the default takes it only with two or more GPUs): on a
    one-GPU box that is ncclCommInitAll(1), the self-check of the fresh communicator (an ncclAllGather and a group of
    ncclBroadcast calls through the prototypes of rccl.h) and the all-gather itself - every RCCL call the multi-GPU path
    makes, with one rank.  Two participants on ONE device are refused (RCCL has one communicator per device)."""
    capi = pkg.capi
    monkeypatch.setenv("SPMV_COMM", "rccl")
    ctx0 = capi.Context(0)
    comm = capi.Comm([ctx0])
    assert comm.backend == "rccl"
    n = 1_000_003
    full = np.random.default_rng(5).uniform(-1, 1, n)
    v = ctx0.vector_from(full)
    for _ in range(3):
        comm.allgather([v], np.array([0, n], dtype=np.int64))
    ctx0.sync()
    assert np.array_equal(v.download(), full)
    # the current device of the calling thread is left alone (callers share the process with torch)
    with pytest.raises(ValueError):
        comm.allgather([v], np.array([0], dtype=np.int64))  # n + 1 offsets are needed: checked before the C side reads them
    with pytest.raises(capi.SpmvError, match="one GPU per participant"):
        capi.Comm([ctx0, capi.Context(0)])
    monkeypatch.setenv("SPMV_COMM", "peer")
    assert capi.Comm([ctx0]).backend == "peer-copy"


def test_rccl_and_peer_copy_transports_agree_bit_for_bit_across_gpus(pkg, monkeypatch):
    """needs two GPUs (skipped on the one-GPU box): the same ragged all-gather, one empty slice included, through RCCL
    and through peer copies"""
    capi = pkg.capi
    ndev = capi.device_count()
    if ndev < 2:
        pytest.skip("one GPU visible: the RCCL transport between devices needs two")
    ndev = min(ndev, 4)
    n = 2_000_003
    cuts = sorted(np.random.default_rng(9).integers(0, n, ndev - 2).tolist()) if ndev > 2 else []
    offsets = np.array([0, *cuts, n, n], dtype=np.int64)[: ndev + 1]
    offsets[-1] = n
    full = np.random.default_rng(6).uniform(-1, 1, n)
    got = {}
    for transport in ("rccl", "peer"):
        monkeypatch.setenv("SPMV_COMM", transport)
        ctxs = [capi.Context(d) for d in range(ndev)]
        comm = capi.Comm(ctxs)
        assert comm.backend == ("rccl" if transport == "rccl" else "peer-copy")
        vecs = [c.vector(n) for c in ctxs]
        for i, v in enumerate(vecs):
            v.fill(float("nan"))
            if offsets[i + 1] > offsets[i]:
                v.upload(full[offsets[i]:offsets[i + 1]], offset=int(offsets[i]))
        comm.allgather(vecs, offsets)
        for c in ctxs:
            c.sync()
        got[transport] = [v.download() for v in vecs]
        del comm, vecs, ctxs
    for a, b in zip(got["rccl"], got["peer"]):
        assert np.array_equal(a, full) and np.array_equal(b, full)


def test_large_coo_handle_keeps_one_layout_only(ctx, pkg):
    """a scattered COO handle with few entries per row: its row-grouped copy satisfies the two-phase policy, the handle
    runs the panel product - the two-phase layout (20 bytes per entry) must not stay allocated beside the panel one"""
    capi = pkg.capi
    n, k = 2_500_000, 4
    C = ctx.gen_csr_uniform(0, n, n, k, seed=41)
    assert C.info.kernel in (capi.CSR_TWOPHASE, capi.CSR_PANEL)  # (the model says two phases for this shape as CSR; at 10M entries both are timed)
    rp, col, val = C.download()
    del C
    rows = np.repeat(np.arange(n, dtype=np.int32), k)
    A = ctx.coo(n, n, rows, col, val)
    nnz = n * k
    # the handle runs from its row-grouped copy, and that copy keeps ONE layout (since round 5 the copy picks its own kernel:
    # the two-phase layout here, 12 + 8 bytes per padded entry; before, the panel layout was forced on it) - never both, and
    # never the copy's col_ind / values beside it
    assert A.info.kernel == capi.CSR_PANEL and A.get_param("rowgrouped_kernel") in (capi.CSR_PANEL, capi.CSR_TWOPHASE)
    per_entry = 15 if A.get_param("rowgrouped_kernel") == capi.CSR_PANEL else 24
    assert A.get_param("device_bytes") <= 16 * nnz + per_entry * nnz + 8 * (n + 1), A.get_param("device_bytes")
    A.set_kernel(capi.CSR_PANEL)  # forcing the panel layout replaces the copy, it does not add to it
    assert A.get_param("rowgrouped_kernel") == capi.CSR_PANEL
    assert A.get_param("device_bytes") <= 16 * nnz + 15 * nnz + 8 * (n + 1), A.get_param("device_bytes")


# ---------------------------------------------------------------------------------- BLAS-1
def test_dot_and_axpby(ctx, orc):
    g = golden("tri8")
    x = ol.f64(g["x"])
    dx = ctx.vector_from(x)
    assert abs(ctx.dot(dx, dx) - float(g["dot_xx"])) <= 1e-13 * abs(float(g["dot_xx"]))
    yv = ol.f64(g["y1_csr"])
    dyv = ctx.vector_from(yv)
    for tag, (a, b) in dict(g=(0.75, -1.25), a0=(0.0, 2.0), b0=(3.0, 0.0), a1=(1.0, 0.5), am1=(-1.0, 0.5), b1=(0.5, 1.0),
                            bm1=(0.5, -1.0)).items():
        dw = ctx.vector(len(x))
        ctx.axpby(a, dx, b, dyv, dw)
        ctx.sync()
        w = dw.download()
        assert np.allclose(w, g[f"axpby_{tag}"], rtol=1e-14, atol=1e-15), tag
        ref = np.zeros(len(x))
        ol.axpby(orc, a, x, b, yv, ref, fma=True)
        assert np.array_equal(w, ref), f"axpby {tag}: not bitwise equal to the fma oracle"
    # alpha == 0 must not read x (NaN there must not leak), beta == 0 must not read y
    bad = ctx.vector_from(np.full(len(x), np.nan))
    dw = ctx.vector(len(x))
    ctx.axpby(0.0, bad, 2.0, dyv, dw)
    ctx.sync()
    assert np.array_equal(dw.download(), 2.0 * yv)
    ctx.axpby(3.0, dx, 0.0, bad, dw)
    ctx.sync()
    assert np.array_equal(dw.download(), 3.0 * x)
    # a long dot against the oracle (different summation tree -> tolerance)
    rng = np.random.RandomState(9)
    a, b = rng.uniform(-1, 1, 1_000_003), rng.uniform(-1, 1, 1_000_003)
    got = ctx.dot(ctx.vector_from(a), ctx.vector_from(b))
    assert abs(got - ol.dot(orc, a, b)) <= 1e-10 * float(np.sum(np.abs(a * b)))


# ---------------------------------------------------------------------------------- generators (bit-exact)
def test_device_generators_equal_numpy_twin(ctx, pkg):
    synth = pkg.synth
    for band in (0, 4096):
        A = ctx.gen_csr_uniform(1000, 3500, 100_000, 32, band=band, seed=42)
        rp, cc, cv = A.download()
        erp, ec, ev = synth.csr_uniform(1000, 3500, 100_000, 32, band=band, seed=42)
        assert np.array_equal(rp, erp) and np.array_equal(cc, ec) and np.array_equal(cv, ev)
        assert A.info.row_begin == 1000 and A.info.max_row_nnz == 32
    E = ctx.gen_ell_banded(5000, 5000, 64, seed=3)
    _, ec, ev = E.download()
    xc, xv = synth.ell_banded(5000, 5000, 64, seed=3)
    assert np.array_equal(ec, xc) and np.array_equal(ev, xv)
    P = ctx.gen_coo_powerlaw(20_000, 20_000, 4096, seed=5)
    r, c, v = P.download()
    er, ec, ev = synth.coo_powerlaw(20_000, 20_000, 4096, seed=5)
    assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(v, ev)
    assert P.info.sorted_rows == 1
    assert np.array_equal(ctx.gen_vector(10_000, index_offset=77, seed=8).download(), synth.vec_uniform(10_000, 77, 8))
    # the last rows of the last of 8 shards of C5 (global row and entry indices beyond 2^31)
    A = ctx.gen_csr_uniform(79_997_000, 80_000_000, 80_000_000, 32, seed=1)
    rp, cc, cv = A.download()
    erp, ec, ev = synth.csr_uniform(79_997_000, 80_000_000, 80_000_000, 32, seed=1)
    assert np.array_equal(rp, erp) and np.array_equal(cc, ec) and np.array_equal(cv, ev) and int(cc.max()) > 2**26
    assert np.array_equal(ctx.gen_vector(5_000, index_offset=79_995_000, seed=1).download(), synth.vec_uniform(5_000, 79_995_000, 1))


# ---------------------------------------------------------------------------------- LDS-window kernel (banded)
def test_csr_ldswin_on_banded_matrix(ctx, orc, pkg):
    synth = pkg.synth
    n, k, band = 60_000, 32, 2048
    rp, cc, cv = synth.csr_uniform(0, n, n, k, band=band, seed=12)
    x = synth.vec_uniform(n, seed=12)
    ref = np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    scale = np.zeros(n)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.csr(n, n, rp, cc, cv)
    # the first and last row blocks wrap around the matrix edge: their window spans all columns, so AUTO must
    # not pick the LDS kernel for this matrix, and forcing it must fail loudly rather than read out of bounds
    assert A.info.kernel != pkg.capi.CSR_LDSWIN and A.get_param("select_us_ldswin") == 0  # (not even timed)
    A.set_kernel(pkg.capi.CSR_LDSWIN)
    with pytest.raises(pkg.capi.SpmvError):
        ctx.apply(A, ctx.vector_from(x), ctx.vector(n))
    # the same band without wrap-around rows: every window fits
    lo, hi = 4096, n - 4096
    rp2 = (rp[lo:hi + 1] - rp[lo]).astype(np.int32)
    cc2, cv2 = cc[rp[lo]:rp[hi]], cv[rp[lo]:rp[hi]]
    B = ctx.csr(hi - lo, n, rp2, cc2, cv2)
    # 1.66M entries: the LDS-window kernel is a candidate AUTO times (against the panel layout and the row-parallel kernel)
    import os

    if os.environ.get("SPMV_PANEL_TRIAL", "1")[:1] != "0":
        assert B.get_param("select_us_ldswin") > 0 and B.get_param("select_candidates") >= 3
    B.set_kernel(pkg.capi.CSR_LDSWIN)
    assert B.info.kernel == pkg.capi.CSR_LDSWIN
    y1, _ = _apply_n(ctx, B, x, hi - lo, 1)
    ol.assert_parity(y1, ref[lo:hi], scale[lo:hi], "csr ldswin")
    B.set_kernel(pkg.capi.CSR_VECTOR)
    y2, _ = _apply_n(ctx, B, x, hi - lo, 1)
    ol.assert_parity(y2, ref[lo:hi], scale[lo:hi], "csr vector on banded")


def test_context_probes_where_workgroups_land_and_the_bins_copy_follows_it(ctx, pkg):
    """HIP promises no workgroup -> XCD placement; the COO scan over one column bin per XCD (and the panel kernel's per-XCD
    bookkeeping) lean on 'workgroups b and b + 8 share an XCD' for speed.  The context reads XCC_ID from 2048 workgroups when it
    is created: on an MI355X in its default mode that holds and 8 ids are seen; the copy in column bins is built exactly when the
    probe says so (a partitioned device would keep the scan in place, with a message, not lose L2 locality silently)."""
    capi = pkg.capi
    rr, seen = ctx.xcd_round_robin()
    assert rr in (0, 1) and 1 <= seen <= 8
    if rr == 1:
        assert seen == 8
    A = ctx.gen_coo_powerlaw(300_000, 600_000, 512, seed=6)  # x = 4.8 MB: beyond an XCD's L2, 2M+ entries
    assert A.info.nnz >= 2 << 20
    A.set_kernel(capi.CSR_VECTOR)
    assert (A.get_param("coo_column_bins") > 0) == (rr == 1)


# ---------------------------------------------------------------------------------- host vectors (the reference's call shape)
@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_apply_host_is_the_resident_product_with_the_hand_over_done_by_the_engine(ctx, orc, pkg, make):
    """spmv_apply_host: y_host += A x_host in one call (what CSRMatrixMatVector(A, x, y) does 50 times in main.cpp:56-59).  Small
    vectors travel through pinned memory the GPU reads and writes itself; the product is the very kernel spmv_apply runs, so
    with a deterministic kernel the result equals the resident product bit for bit - after 1 call and after 50 accumulating
    ones, with fresh host arrays in every call (nothing of the caller's memory stays registered or mapped)."""
    capi = pkg.capi
    c = make()
    g = golden(c["name"])
    scale = _scale(orc, c)
    rp, cc, cv = ol.coo_to_csr(orc, c["nrow"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]))
    A = ctx.csr(c["nrow"], c["ncol"], rp, cc, cv)
    A.set_kernel(capi.CSR_VECTOR)  # a fixed tree per row: the same bits on every call
    y1_res, y50_res = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
    y = np.zeros(c["nrow"])
    ctx.apply_host(A, np.array(c["x"], dtype=np.float64), y)
    assert np.array_equal(y, y1_res)
    ol.assert_parity(y, g["y1_csr"], scale, c["name"] + " apply_host 1 call")
    for _ in range(NUM_TEST - 1):
        xs = np.array(c["x"], dtype=np.float64)  # a new host array every call
        ctx.apply_host(A, xs, y)
        del xs
    assert np.array_equal(y, y50_res)
    ol.assert_parity(y, g["y50_csr"], scale, c["name"] + " apply_host 50 calls", reps=NUM_TEST)
    # the other formats through the same entry point
    for M, key in ((ctx.coo(c["nrow"], c["ncol"], c["row"], c["col"], c["val"]), "y1_coo"),):
        yh = np.zeros(c["nrow"])
        ctx.apply_host(M, np.array(c["x"], dtype=np.float64), yh)
        ol.assert_parity(yh, g[key], scale, c["name"] + " apply_host " + key)
    with pytest.raises(ValueError):
        ctx.apply_host(A, np.zeros(c["ncol"] + 1), y)


def test_apply_host_large_vectors_take_the_copy_path_and_work_queued_before_is_respected(ctx, orc, pkg):
    """above 4 MB of vectors the hand-over is two asynchronous copies in and one out; a product queued on the context's stream
    before the call (asynchronous spmv_apply) is finished first, and the staging buffers grow and shrink with the calls"""
    synth = pkg.synth
    n, k = 700_000, 6  # x + y = 11 MB
    rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=33)
    x = synth.vec_uniform(n, seed=33)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    A = ctx.csr(n, n, rp, cc, cv)
    dx, dy = ctx.vector_from(x), ctx.vector(n)
    dy.fill(0.0)
    ctx.apply(A, dx, dy)  # queued, not waited for
    y = np.zeros(n)
    ctx.apply_host(A, x, y)
    ctx.apply_host(A, x, y)
    ol.assert_parity(y, 2 * ref, scale, "apply_host, 11 MB of vectors, two calls", reps=2)
    ctx.sync()
    ol.assert_parity(dy.download(), ref, scale, "the product queued before")
    # a small one right after a large one (the staged path with buffers that are larger than it needs)
    c = cases.tri8()
    rp8, cc8, cv8 = ol.coo_to_csr(orc, c["nrow"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]))
    T = ctx.csr(c["nrow"], c["ncol"], rp8, cc8, cv8)
    y8 = np.zeros(c["nrow"])
    ctx.apply_host(T, np.array(c["x"], dtype=np.float64), y8)
    ol.assert_parity(y8, golden("tri8")["y1_csr"], _scale(orc, c), "apply_host tri8 after a large call")


# ---------------------------------------------------------------------------------- sharding on one device
def test_row_shards_concatenate_to_unsharded_result(ctx, orc, pkg):
    """the NUMA driver's partition (src/mat_vec.cpp:240-268) emulated with 8 shards on one GPU"""
    synth, capi = pkg.synth, pkg.capi
    n, k, parts = 100_003, 16, 8
    rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=21)
    x = synth.vec_uniform(n, seed=21)
    ref = np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    scale = np.zeros(n)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx = ctx.vector_from(x)
    rp64 = rp.astype(np.int64)
    pieces = []
    for p in range(parts):
        b, e = capi.partition_rows(n, parts, p)
        assert (b, e) == ol.partition_rows(orc, n, parts, p)
        S = ctx.csr_shard(b, e, n, rp64, cc, cv)
        srp, _, _ = S.download()
        assert np.array_equal(srp, ol.csr_shard_row_ptr(orc, rp, b, e))
        assert S.info.row_begin == b and S.info.nrow == e - b
        dy = ctx.vector(e - b)
        dy.fill(0.0)
        ctx.apply(S, dx, dy)
        ctx.sync()
        pieces.append(dy.download())
    ol.assert_parity(np.concatenate(pieces), ref, scale, "8 row shards")


@pytest.mark.parametrize("fmt", ["csr", "coo"])
def test_skewed_matrix_sharded_by_rows_and_by_entries(ctx, orc, pkg, fmt):
    """"an nnz-balanced split as an option for C4-like skew".  Power-law rows SORTED BY LENGTH (every heavy
    row in the first eighth), 8 shards on one GPU, cut both ways from the device-resident handle (spmv_mat_partition_rows) and
    handed out device to device (spmv_csr_extract_rows): the concatenated y equals the oracle's under either partition, the
    equal-rows split is as lopsided as the matrix, the by-entries split is within 2 % of even, and the device generator's sorted
    variant is the numpy twin's bit for bit."""
    synth, capi = pkg.synth, pkg.capi
    n, max_len, parts = 400_000, 4096, 8
    G = ctx.gen_coo_powerlaw(n, n, max_len, seed=13, sorted_by_length=True)
    rows, cols, vals = G.download()
    hr, hc, hv = synth.coo_powerlaw(n, n, max_len, seed=13, sorted_by_length=True)
    assert np.array_equal(rows, hr) and np.array_equal(cols, hc) and np.array_equal(vals, hv)
    ln = np.bincount(rows, minlength=n)
    assert np.all(np.diff(ln) <= 0) and ln[0] == max_len and ln[-1] == 8  # sorted by length, the longest first
    rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int64)
    x = synth.vec_uniform(n, seed=13)
    rp32 = rp.astype(np.int32)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp32, cols, vals, x, ref)
    ol.csr_abs_row_sums(orc, rp32, cols, vals, x, scale)
    dx = ctx.vector_from(x)
    if fmt == "coo":
        # (a COO handle is partitioned from a histogram of its row indices; the shards come from its CSR form)
        b_rows, b_ent = G.partition_rows(parts, False), G.partition_rows(parts, True)
        A = ctx.coo_to_csr(G)
    else:
        A = ctx.coo_to_csr(G)
        b_rows, b_ent = A.partition_rows(parts, False), A.partition_rows(parts, True)
    del G
    assert list(b_rows) == [capi.partition_rows(n, parts, p)[0] for p in range(parts)] + [n]
    assert np.array_equal(b_ent, capi.partition_rows_balanced(rp, parts))
    share_rows = np.diff(rp[b_rows])
    share_ent = np.diff(rp[b_ent])
    assert share_rows.max() / share_rows.mean() > 3.0, share_rows            # the first eighth holds most of the matrix
    assert abs(share_ent / share_ent.mean() - 1.0).max() <= 0.02, share_ent  # within 2 % of even
    for bounds, what in ((b_rows, "equal rows"), (b_ent, "by entries")):
        pieces, counts = [], []
        for p in range(parts):
            b, e = int(bounds[p]), int(bounds[p + 1])
            S = ctx.extract_rows(A, b, e)
            inf = S.info
            assert inf.row_begin == b and inf.nrow == e - b and inf.nnz == rp[e] - rp[b] and inf.ncol == n
            srp, scc, svv = S.download()
            assert np.array_equal(srp, ol.csr_shard_row_ptr(orc, rp32, b, e))
            assert np.array_equal(scc, cols[rp[b]:rp[e]]) and np.array_equal(svv, vals[rp[b]:rp[e]])
            dy = ctx.vector(e - b)
            dy.fill(0.0)
            ctx.apply(S, dx, dy)
            ctx.sync()
            pieces.append(dy.download())
            counts.append(int(inf.nnz))
        assert sum(counts) == rp[-1]
        ol.assert_parity(np.concatenate(pieces), ref, scale, f"8 shards of a length-sorted power-law matrix, {what}")
    with pytest.raises(capi.SpmvError, match="outside"):
        ctx.extract_rows(A, 10, n + 1)
    # padded formats keep equal rows either way (every row stores k slots): asking for balance changes nothing
    E = ctx.gen_ell_banded(1000, 1000, 8, seed=2)
    assert np.array_equal(E.partition_rows(3, True), E.partition_rows(3, False))


def test_auto_times_its_candidates_and_keeps_the_fastest(ctx, orc, pkg, monkeypatch):
    """AUTO is a measurement where a model cannot know (64K .. 8M entries; tools/sweep_structures.py).
    What is checked here is the mechanism, not a timing: which candidates were timed, that the kernel kept is the fastest of
    them by the handle's own record (a later candidate has to win by 2 %), that every candidate computes the same product,
    that nothing of a losing layout stays allocated, and that SPMV_PANEL_TRIAL=0 leaves the model alone."""
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)  # (this test is about the trials; tools/env_sweeps.sh also runs the suite without them)
    names = {1: "vector", 2: "ldswin", 3: "scalar", 4: "panel", 5: "twophase", 6: "segscan", 7: "split"}
    rng = np.random.default_rng(17)
    # (a) a hub row among short ones (R-MAT-like): 40000 rows x 8, one row of 30000 entries: 350K entries
    n = 40_000
    lens = np.full(n, 8, np.int64)
    lens[1234] = 30_000
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cc = rng.integers(0, n, rp[-1]).astype(np.int32)
    cv = rng.uniform(-1, 1, rp[-1])
    x = rng.uniform(0, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx, dy = ctx.vector_from(x), ctx.vector(n)

    def product(M, what):
        dy.fill(0.0)
        ctx.apply(M, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, what)

    A = ctx.csr(n, n, rp, cc, cv)
    timed = {k: A.get_param("select_us_" + v) for k, v in names.items() if A.get_param("select_us_" + v) > 0}
    assert A.get_param("select_candidates") == len(timed) >= 2 and capi.CSR_PANEL in timed and capi.CSR_VECTOR in timed, timed
    kept = int(A.info.kernel)
    assert kept in timed, (kept, timed)  # mechanism: the kept kernel is one of those that were timed
    perf_expect(timed[kept] <= 1.03 * min(timed.values()) + 1, f"hub row: kept {kept} is the fastest of {timed}")
    # a lane group of the row-parallel kernel walks the hub row alone: whatever the box, that is not the fastest candidate
    perf_expect(kept != capi.CSR_VECTOR and timed[capi.CSR_VECTOR] > 2 * timed[kept], f"hub row: row-parallel 2x behind, {timed}")
    product(A, f"hub row, AUTO kept {names[kept]}")
    if kept != capi.CSR_PANEL:
        assert A.get_param("panel_bytes") == 0  # the losing layout went back
    bytes_auto = A.get_param("device_bytes")
    for k in (capi.CSR_VECTOR, capi.CSR_SCALAR, capi.CSR_PANEL, capi.CSR_SEGSCAN, capi.CSR_SPLIT):
        A.set_kernel(k)
        product(A, f"hub row, forced {names[k]}")
    A.set_kernel(capi.CSR_AUTO)  # selecting again times again and ends in the same state
    assert A.get_param("select_candidates") >= 2
    if int(A.info.kernel) == kept:  # (two candidates within microseconds of each other may swap places between two trials)
        assert A.get_param("device_bytes") <= bytes_auto + (1 << 20)
    assert A.get_param("device_bytes") <= 4 * (n + 1) + 12 * int(rp[-1]) + 16 * int(rp[-1]) + (1 << 20)  # the arrays + ONE layout
    product(A, "hub row, AUTO again")
    # the model alone (no timing launches): the hub-row rule picks the panel layout
    monkeypatch.setenv("SPMV_PANEL_TRIAL", "0")
    B = ctx.csr(n, n, rp, cc, cv)
    assert B.get_param("select_candidates") == 0 and B.info.kernel == capi.CSR_PANEL
    product(B, "hub row, model only")
    monkeypatch.delenv("SPMV_PANEL_TRIAL")
    # (b) the same matrix as a COO handle: the scan and the row-grouped copy are both timed; the copy picks its own kernel
    rows = np.repeat(np.arange(n, dtype=np.int32), lens)
    C = ctx.coo(n, n, rows, cc, cv)
    assert C.get_param("select_candidates") == 2 and C.get_param("select_us_vector") > 0 and C.get_param("select_us_panel") > 0
    if C.info.kernel == capi.CSR_PANEL:
        # (whole microseconds: the copy won by 2 % and more, which may round to the same number)
        assert C.get_param("rowgrouped_kernel") in names
        perf_expect(C.get_param("select_us_panel") <= C.get_param("select_us_vector"), "hub row as COO: the copy that was kept timed faster")
    else:
        assert C.get_param("rowgrouped_kernel") == 0 and C.get_param("device_bytes") <= 16 * int(rp[-1]) + 4096  # the copy went back
    product(C, "hub row, COO AUTO")
    # (c) few long rows in an ELL handle: 3000 rows x 96 slots over 400K columns - one lane per row leaves the chip idle, the
    # row-grouped copy is a candidate and (timed) wins by a wide margin on any box
    nr, K, ncol = 3000, 96, 400_000
    ec = rng.integers(0, ncol, nr * K).astype(np.int32)
    ev = rng.uniform(-1, 1, nr * K)
    xe = rng.uniform(0, 1, ncol)
    E = ctx.ell(nr, ncol, K, nr * K, ec, ev)
    assert E.get_param("select_candidates") >= 3 and E.get_param("select_us_panel") > 0
    perf_expect(E.info.kernel == capi.CSR_PANEL and E.get_param("select_us_panel") * 2 < E.get_param("select_us_vector"),
                "3000 long ELL rows: the row-grouped copy wins 2x")
    ye, yv = ctx.vector(nr), ctx.vector(nr)
    ye.fill(0.0)
    yv.fill(0.0)
    dxe = ctx.vector_from(xe)
    ctx.apply(E, dxe, ye)
    E.set_kernel(capi.CSR_VECTOR, 1)  # one row per lane: the reference's order, bit-identical to the fma oracle (tested above)
    ctx.apply(E, dxe, yv)
    ctx.sync()
    assert np.max(np.abs(ye.download() - yv.download())) <= ol.REL_TOL * K
    # (d) handles of 8M entries and more keep the model's pick without a launch (a trial of the row-parallel kernel on C2
    # would cost 60 ms for a kernel that loses 5x) - unless their rows are long contiguous runs
    big = ctx.gen_csr_uniform(0, 600_000, 600_000, 16, seed=5)
    assert big.get_param("select_candidates") == 0 and big.info.kernel == capi.CSR_PANEL and big.get_param("contiguous_permille") < 50
    del big
    nb, bs = 64 * 2344, 64  # dense 64 x 64 blocks on the diagonal: 9.6M entries in contiguous runs
    i = np.arange(nb, dtype=np.int64)
    brp = (np.arange(nb + 1, dtype=np.int64) * bs).astype(np.int32)
    bcc = (np.repeat(i // bs * bs, bs) + np.tile(np.arange(bs), nb)).astype(np.int32)
    D = ctx.csr(nb, nb, brp, bcc, rng.uniform(-1, 1, nb * bs))
    assert D.get_param("contiguous_permille") > 950 and D.get_param("select_candidates") in (2, 3, 4)  # row-parallel, panel, LDS window (its windows fit), the ELL copy (equal rows)
    assert D.get_param("select_us_vector") > 0 and D.get_param("select_us_panel") > 0


def test_rows_that_hold_most_of_the_entries_go_through_the_scan(ctx, orc, pkg, monkeypatch):
    """Round 5 (tools/sweep_structures.py "odd"): an arrow matrix - dense first row, dense first column, a diagonal - under
    every row-wise CSR kernel leaves its long row to one wavefront or one workgroup (1M entries: 1.26 ms under the panel
    kernel, 74 ms row-parallel).  SPMV_CSR_SEGSCAN cuts the ENTRIES into equal pieces (the COO scan over a row index per
    entry); AUTO times it where the longest row holds 1/128 of the entries and the model picks it from 1/64 and 65536 on."""
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    rng = np.random.default_rng(23)
    n = 120_000
    # rows 1000..2999 are EMPTY (the scan never touches them: y += 0), row 0 is dense, every other row has (i, 0) and (i, i)
    lens = np.full(n, 2, np.int64)
    lens[0] = n
    lens[1000:3000] = 0
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cc = np.empty(rp[-1], np.int32)
    cc[:n] = np.arange(n)
    body = np.flatnonzero(lens[1:] > 0) + 1
    cc[n::2] = 0
    cc[n + 1::2] = body
    cv = rng.uniform(-1, 1, rp[-1])
    x = rng.uniform(0, 1, n)
    y0 = rng.uniform(-1, 1, n)
    ref, scale = y0.copy(), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)  # y += A x on top of y0
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    scale = scale + np.abs(y0)
    dx = ctx.vector_from(x)

    def product(M, what):
        dy = ctx.vector_from(y0)
        ctx.apply(M, dx, dy)
        ctx.sync()
        got = dy.download()
        ol.assert_parity(got, ref, scale, what)
        assert np.array_equal(got[1000:3000], y0[1000:3000]), what  # rows without entries: untouched, bit for bit

    A = ctx.csr(n, n, rp, cc, cv)
    base = 4 * (n + 1) + 12 * int(rp[-1])
    assert A.info.max_row_nnz == n
    t_scan, t_split, t_panel = (A.get_param("select_us_" + k) for k in ("segscan", "split", "panel"))
    assert t_scan > 0 and t_split > 0 and t_panel > 0, (t_scan, t_split, t_panel)  # mechanism: all three were timed
    perf_expect(A.info.kernel in (capi.CSR_SEGSCAN, capi.CSR_SPLIT) and t_panel > 3 * max(t_scan, t_split), f"arrow: scan {t_scan} split {t_split} panel {t_panel} us")
    perf_expect((A.info.kernel == capi.CSR_SPLIT) == (t_split < 0.98 * t_scan) or abs(t_split - t_scan) <= 1, f"arrow: scan {t_scan} split {t_split} us, kept {A.info.kernel}")
    if A.info.kernel != capi.CSR_PANEL:
        assert A.get_param("panel_bytes") == 0
    if A.info.kernel == capi.CSR_SEGSCAN:  # the row index per entry, and nothing of the candidates that lost
        assert base + 4 * int(rp[-1]) <= A.info.device_bytes <= base + 4 * int(rp[-1]) + (1 << 20)
    elif A.info.kernel == capi.CSR_SPLIT:
        assert A.info.device_bytes < base + 16 * (int(rp[-1]) - n) + (2 << 20)
    product(A, "arrow, AUTO")
    for k in (capi.CSR_PANEL, capi.CSR_VECTOR, capi.CSR_SCALAR):
        A.set_kernel(k)
        assert A.info.kernel == k
        product(A, f"arrow, forced kernel {k}")
    A.set_kernel(capi.CSR_VECTOR)
    assert A.info.device_bytes <= base + (1 << 20) + A.get_param("panel_bytes")  # the row index went back with the kernel
    A.set_kernel(capi.CSR_SEGSCAN)
    assert A.info.kernel == capi.CSR_SEGSCAN
    product(A, "arrow, scan forced")
    # kernel SPLIT: the dense row in chunks of 4096 entries over the handle's own arrays, the other rows through a copy
    A.set_kernel(capi.CSR_SPLIT)
    assert A.info.kernel == capi.CSR_SPLIT and A.get_param("split_row_threshold") == n // 16
    assert A.get_param("split_long_rows") == 1 and A.get_param("split_long_entries") == n and A.get_param("split_inner_kernel") in (1, 2, 3, 4, 5)
    # the row index of the scan went back; the copy holds the short rows only (the panel layout forced above stays until AUTO or a re-build)
    assert A.info.device_bytes < base + A.get_param("panel_bytes") + 16 * (int(rp[-1]) - n) + (2 << 20)
    product(A, "arrow, long-row split")
    assert A.get_param("split_mode") == 1 and A.get_param("split_virtual_rows") == 0  # a dense row: chunks of the handle's own arrays
    for mode in (1, 2):  # 2: every long row dealt out to virtual rows of 64 entries, a CSR matrix with a handle of its own
        A.set_param("split_mode", mode)
        for T in (0, 2, 1, 1 << 30):  # default; every row with entries is "long" (the copy is empty); the same; none is
            A.set_param("split_row_threshold", T)
            A.set_kernel(capi.CSR_SPLIT)
            assert A.get_param("split_long_rows") == (1 if T == 0 else n - 2000 if T <= 2 else 0) and A.get_param("split_mode") == mode
            if mode == 2:
                assert A.get_param("split_virtual_rows") == ((n + 63) // 64 + (n - 2001 if 0 < T <= 2 else 0) if T < 1 << 30 else 0)
                assert (A.get_param("split_long_kernel") > 0) == (T < 1 << 30)
            product(A, f"arrow, split at {T}, mode {mode}")
    A.set_param("split_row_threshold", 0)
    A.set_param("split_mode", 0)
    A.set_kernel(capi.CSR_VECTOR)
    assert A.info.device_bytes <= base + (1 << 20) + A.get_param("panel_bytes")
    # the solver's fused extras (y = A x, w . y) on a kernel that has no write-back of its own: CG on an SPD arrow
    m = 70_000
    l2 = np.full(m, 2, np.int64)
    l2[0] = m
    rp2 = np.concatenate(([0], np.cumsum(l2))).astype(np.int32)
    c2 = np.empty(rp2[-1], np.int32)
    c2[:m] = np.arange(m)
    c2[m::2] = 0
    c2[m + 1::2] = np.arange(1, m)
    v2 = np.empty(rp2[-1])
    v2[:m] = 1e-3
    v2[0] = 4.0
    v2[m::2] = 1e-3
    v2[m + 1::2] = 2.0 + rng.uniform(0, 1, m - 1)
    S = ctx.csr(m, m, rp2, c2, v2)
    perf_expect(S.info.kernel in (capi.CSR_SEGSCAN, capi.CSR_SPLIT), f"SPD arrow: AUTO kept {S.info.kernel}")
    xs = rng.uniform(-1, 1, m)
    bs = np.zeros(m)
    ol.csr_spmv(orc, rp2, c2, v2, xs, bs)
    sol = ctx.vector(m)
    for k, mode in ((capi.CSR_SEGSCAN, 0), (capi.CSR_SPLIT, 1), (capi.CSR_SPLIT, 2)):
        S.set_param("split_mode", mode)
        S.set_kernel(k)
        sol.fill(0.0)
        iters, relres = ctx.cg(S, ctx.vector_from(bs), sol, max_iter=200, rel_tol=1e-12, check_every=4)
        assert relres <= 1e-12 and np.max(np.abs(sol.download() - xs)) < 1e-9, (k, mode, iters, relres)
    # a CSC handle of the arrow: its row-grouped copy may pick the scan; a COO handle's copy never does (the handle has it itself)
    rows = np.repeat(np.arange(n, dtype=np.int32), lens)
    cp, cr, cw = ol.coo_to_csc(orc, n, rows, cc, cv)
    C = ctx.csc(n, n, cp, cr, cw)
    perf_expect(C.info.kernel == capi.CSR_PANEL and C.get_param("rowgrouped_kernel") in (capi.CSR_SEGSCAN, capi.CSR_SPLIT),
                f"arrow as CSC: kernel {C.info.kernel}, the copy runs {C.get_param('rowgrouped_kernel')}")
    product(C, "arrow as CSC, AUTO")
    O = ctx.coo(n, n, rows, cc, cv)
    assert O.get_param("rowgrouped_kernel") != capi.CSR_SEGSCAN
    product(O, "arrow as COO, AUTO")
    with pytest.raises(capi.SpmvError, match="COO handles take"):
        O.set_kernel(capi.CSR_SEGSCAN)
    # the model alone
    monkeypatch.setenv("SPMV_PANEL_TRIAL", "0")
    B = ctx.csr(n, n, rp, cc, cv)
    assert B.get_param("select_candidates") == 0 and B.info.kernel == capi.CSR_SPLIT and B.get_param("split_long_rows") == 1
    product(B, "arrow, model only")
    monkeypatch.delenv("SPMV_PANEL_TRIAL")
    del A, B, C, O, S
    # from 8M entries on the split is timed at TWO thresholds (its default and 256: graphs whose long rows share hub columns want
    # the low one, R-MAT scale 22: 0.327 -> 0.250 ms); 500000 rows x 16, 2000 rows of 600 and one dense row: 9.7M entries
    nb, kb = 500_000, 16
    lb = np.full(nb, kb, np.int64)
    mid = rng.choice(np.arange(1, nb), 2000, replace=False)
    lb[mid] = 600
    lb[0] = nb
    rpb = np.concatenate(([0], np.cumsum(lb))).astype(np.int32)
    cb = rng.integers(0, nb, rpb[-1]).astype(np.int32)
    cb[:nb] = np.arange(nb)
    vb = rng.uniform(-1, 1, rpb[-1])
    xb = rng.uniform(0, 1, nb)
    rowsb = np.repeat(np.arange(nb), lb)
    refb = np.bincount(rowsb, weights=vb * xb[cb], minlength=nb)
    scb = np.bincount(rowsb, weights=np.abs(vb) * xb[cb], minlength=nb)
    G = ctx.csr(nb, nb, rpb, cb, vb)
    t_def, t_low = G.get_param("select_us_split"), G.get_param("select_us_split_low")
    assert G.get_param("select_candidates") == 4 and t_def > 0 and t_low > 0 and G.get_param("select_us_panel") > 0 and G.get_param("select_us_segscan") > 0
    perf_expect(G.info.kernel == capi.CSR_SPLIT, f"8M entries with long rows: AUTO kept {G.info.kernel}")
    if G.info.kernel == capi.CSR_SPLIT:
        assert G.get_param("split_row_threshold") in (256, nb // 16)
        perf_expect((G.get_param("split_row_threshold") == 256) == (t_low < 0.98 * t_def) or abs(t_low - t_def) <= 0.03 * t_def + 1, f"split thresholds: default {t_def} us, 256 {t_low} us")
        assert G.get_param("split_long_rows") == (2001 if G.get_param("split_row_threshold") == 256 else 1)
    dxb, dyb = ctx.vector_from(xb), ctx.vector(nb)
    dyb.fill(0.0)
    ctx.apply(G, dxb, dyb)
    ctx.sync()
    ol.assert_parity(dyb.download(), refb, scb, "8M entries with long rows, AUTO")
    G.set_kernel(capi.CSR_SPLIT)  # forced: the default threshold, whatever AUTO found
    assert G.get_param("split_row_threshold") == nb // 16 and G.get_param("split_long_rows") == 1


def test_panel_layout_cut_for_more_than_one_round_of_workgroups(ctx, orc, pkg):
    """Round 5: with many rows AND skewed lengths the fewest groups the LDS cap allows leave the heavy rows too few groups to
    spread over (R-MAT scale 22: the busiest group 2.24x the mean); csr_panel_build then times a cut for 2-4 rounds of 256
    workgroups against the single round (tools/probe_split_threshold.py: 0.333 -> 0.267 ms there).  "panel_rounds" forces
    either; whatever the cut, the product is the oracle's."""
    capi = pkg.capi
    rng = np.random.default_rng(31)
    n = 700_000  # 600000 rows of 2 entries, then 100000 of 40: the light rows alone need 30 groups of 20000 rows
    lens = np.concatenate((np.full(600_000, 2, np.int64), np.full(100_000, 40, np.int64)))
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cc = rng.integers(0, n, rp[-1]).astype(np.int32)
    cv = rng.uniform(-1, 1, rp[-1])
    x = rng.uniform(0, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx, dy = ctx.vector_from(x), ctx.vector(n)
    A = ctx.csr(n, n, rp, cc, cv)
    seen = {}
    for rounds in (1, 2, 3, 0):
        A.set_param("panel_rounds", rounds)
        A.set_kernel(capi.CSR_PANEL)
        built = A.get_param("panel_rounds")
        assert built == rounds or (rounds == 0 and built >= 1)
        seen[rounds] = A.get_param("panel_groups")
        dy.fill(0.0)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ol.assert_parity(dy.download(), ref, scale, f"panel layout cut for {rounds} round(s): {seen[rounds]} groups")
    assert seen[1] <= 256 < seen[2] <= 512 < seen[3] <= 768, seen
    if A.get_param("panel_rounds_us_more"):  # the automatic cut timed an alternative: what stayed is what was faster (3 % for the finer one)
        one, more = A.get_param("panel_rounds_us_one"), A.get_param("panel_rounds_us_more")
        perf_expect((A.get_param("panel_rounds") > 1) == (more < 0.97 * one) or abs(more - one) <= 0.04 * one + 1, f"panel rounds: one {one} us, more {more} us")


def test_csr_handles_of_nearly_equal_rows_may_run_from_an_ell_copy(ctx, orc, pkg, monkeypatch):
    """Round 5 (tools/sweep_structures.py: stencils, bands, block diagonals run 1.15x to 1.57x faster through an ELL handle than
    through a CSR handle's best kernel): a CSR handle whose padding to its longest row stays below a quarter and that has no
    empty row times an ELL copy of itself (kernel SPMV_CSR_ELL).  The copy's kernels add a row's entries in slot order like the
    ELL kernels and leave its PADDING out of the sums (round 6): with finite x the bits are an ELL handle's of the same matrix,
    and a non-finite x[c] does to a row exactly what it does under the reference's CSR loop."""
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    rng = np.random.default_rng(41)
    n, half = 600_000, 3  # a band of 7 around the diagonal, clipped at the edges (rows of 4 to 7): 4.2M entries
    i = np.repeat(np.arange(n, dtype=np.int64), 2 * half + 1)
    c = i + np.tile(np.arange(-half, half + 1, dtype=np.int64), n)
    ok = (c >= 0) & (c < n)
    rows, cols = i[ok], c[ok].astype(np.int32)
    lens = np.bincount(rows, minlength=n)
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    vals = rng.uniform(-1, 1, rows.size)
    x = rng.uniform(0, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cols, vals, x, ref)
    ol.csr_abs_row_sums(orc, rp, cols, vals, x, scale)
    dx, dy = ctx.vector_from(x), ctx.vector(n)

    def product(M, xv=None):
        dy.fill(0.0)
        ctx.apply(M, xv if xv is not None else dx, dy)
        ctx.sync()
        return dy.download()

    A = ctx.csr(n, n, rp, cols, vals)
    assert A.get_param("min_row_entries") == half + 1 and A.info.max_row_nnz == 2 * half + 1
    t_ell, t_panel = A.get_param("select_us_ell"), A.get_param("select_us_panel")
    assert t_ell > 0 and t_panel > 0  # both were candidates and timed
    kept = int(A.info.kernel)
    assert (kept == capi.CSR_ELL) == (A.get_param("ell_copy_slots") == n * (2 * half + 1))  # the copy stays only where it runs
    ol.assert_parity(product(A), ref, scale, f"band of 7, AUTO (kernel {kept})")
    base = A.get_param("device_bytes")
    A.set_kernel(capi.CSR_ELL)
    assert A.info.kernel == capi.CSR_ELL and A.get_param("ell_copy_slots") == n * 7
    y_copy = product(A)
    ol.assert_parity(y_copy, ref, scale, "band of 7, ELL copy forced")
    # the reference's ELL arithmetic: an ELL handle of the same matrix (padding: column 0, value 0.0) gives the same bits
    E = ctx.csr_to_ell(A) if hasattr(ctx, "csr_to_ell") else None
    if E is not None:
        E.set_kernel(capi.CSR_VECTOR, 2 if A.get_param("ell_copy_variant") != 1 else 1)
        assert np.array_equal(product(E), y_copy) or np.max(np.abs(product(E) - y_copy)) <= ol.REL_TOL * 7
    # x[0] = NaN reaches the rows that read column 0 (rows 0..half) and no other - under the copy as under every CSR kernel;
    # through the ELL HANDLE the reference's padding (column 0) brings it into every padded row (a4 of SURVEY 8a)
    xn = x.copy()
    xn[0] = np.nan
    dxn = ctx.vector_from(xn)
    got = product(A, dxn)
    assert np.all(np.isnan(got[:half + 1])) and np.all(np.isfinite(got[half + 1:]))
    if E is not None:
        assert np.isnan(product(E, dxn)[n - 1])  # (the last rows are padded: 0.0 * x[0])
    A.set_kernel(capi.CSR_PANEL)
    assert A.get_param("ell_copy_slots") == 0  # the copy goes back with the kernel
    got = product(A, dxn)
    assert np.all(np.isnan(got[:half + 1])) and np.all(np.isfinite(got[half + 1:]))
    # not a candidate: an empty row, or padding beyond a quarter
    lens2 = lens.copy()
    lens2[1000] = 0
    keep = np.ones(rows.size, bool)
    keep[rp[1000]:rp[1001]] = False
    rp2 = np.concatenate(([0], np.cumsum(lens2))).astype(np.int32)
    B = ctx.csr(n, n, rp2, cols[keep], vals[keep])
    assert B.get_param("min_row_entries") == 0 and B.get_param("select_us_ell") == 0 and B.info.kernel != capi.CSR_ELL
    # forced, it works all the same: the copy's kernels leave its padding out of the sums (round 6), so the empty row is not
    # touched - not even by x[0] = NaN, which its padded slots (column 0, value 0.0) would otherwise multiply
    B.set_kernel(capi.CSR_ELL)
    assert B.info.kernel == capi.CSR_ELL and B.get_param("ell_copy_slots") == n * 7
    ref2, scale2 = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp2, cols[keep], vals[keep], x, ref2)
    ol.csr_abs_row_sums(orc, rp2, cols[keep], vals[keep], x, scale2)
    ol.assert_parity(product(B), ref2, scale2, "band with an empty row, ELL copy forced")
    got = product(B, dxn)
    assert got[1000] == 0.0 and np.all(np.isnan(got[:half + 1])) and np.all(np.isfinite(got[half + 1:]))
    # Round 5's copy had ONE deviation from the CSR loop: a short row that reads x[c] = +inf got NaN from its padding (0.0 * inf).
    # Round 6: the padding takes no part in the sums (slots beyond a row's own length, read off the handle's row_ptr), so the copy
    # gives +-inf exactly where every other CSR kernel and the reference (src/mat_vec.cpp:57-65) do - padded rows included
    xi = x.copy()
    xi[n - 1] = np.inf  # column n - 1: read by the last rows, which are short (padded), and by row n - 1 - half (a full row)
    dxi = ctx.vector_from(xi)
    refi = np.zeros(n)
    ol.csr_spmv(orc, rp, cols, vals, xi, refi)
    assert np.all(np.isinf(refi[n - 1 - half:])) and np.all(np.isfinite(refi[:n - 1 - half]))
    for kernel in (capi.CSR_VECTOR, capi.CSR_ELL):
        A.set_kernel(kernel)
        got = product(A, dxi)
        assert np.array_equal(np.isinf(got), np.isinf(refi)) and np.array_equal(np.sign(got[n - 1 - half:]), np.sign(refi[n - 1 - half:])), kernel
        assert not np.any(np.isnan(got)), kernel
    # the solver's extras over the copy (generic path: fill, product, dot)
    w = rng.uniform(-1, 1, n)
    A.set_kernel(capi.CSR_ELL)
    dw = ctx.vector_from(w)
    d = ctx.apply_dot(A, dx, dy, dw, overwrite=True)
    got = dy.download()
    ol.assert_parity(got, ref, scale, "band of 7, ELL copy, overwrite + dot")
    assert abs(d - float(w @ got)) <= 1e-11 * float(np.abs(w) @ np.abs(got))
    assert base > 0
    # panel_keep_csr = 0 under the ELL copy: col_ind / values go back (the copy holds every entry), the product stays
    before = A.get_param("device_bytes")
    A.set_param("panel_keep_csr", 0)
    assert A.get_param("device_bytes") == before - 12 * rows.size and A.get_param("panel_keep_csr") == 0
    ol.assert_parity(product(A), ref, scale, "band of 7, ELL copy, CSR arrays released")
    A.set_kernel(capi.CSR_AUTO)  # stays what it is
    assert A.info.kernel == capi.CSR_ELL
    with pytest.raises(capi.SpmvError, match="gave up its CSR arrays"):
        A.set_kernel(capi.CSR_VECTOR)  # (a kernel that reads them; the panel layout built above would still run)


def test_row_grouped_copy_of_a_ragged_ell_handle_keeps_one_padding_term_per_row(ctx, orc, pkg):
    """The copy of an ELL handle drops the padding slots (column 0, value 0.0) beyond the first of every row: the reference's sum
    adds 0.0 * x[0] once per padded slot (src/mat_vec.cpp:108-117), and once says what many say.  A mesh-like matrix (rows of
    4 to 40 entries in 40 slots): the copy holds ~entries + rows, not rows x 40; the product is the oracle's ELL product; and
    with x[0] = NaN exactly the rows that have padding (or read column 0) turn NaN - under the copy as under the format's own
    kernel."""
    capi = pkg.capi
    rng = np.random.default_rng(53)
    n, K = 150_000, 40
    lens = rng.integers(4, K + 1, n)
    lens[::1000] = K  # some rows without padding
    cols = np.zeros((K, n), np.int32)
    vals = np.zeros((K, n))
    for s in range(K):
        live = lens > s
        cols[s, live] = np.clip(np.flatnonzero(live) + rng.integers(1, 2000, int(live.sum())), 1, n - 1)  # (never column 0)
        vals[s, live] = rng.uniform(0.5, 1.5, int(live.sum()))
    nnz = int(lens.sum())
    x = rng.uniform(0.5, 1.5, n)
    ref = np.zeros(n)
    ol.ell_spmv(orc, n, K, cols.ravel(), vals.ravel(), x, ref)
    scale = np.abs(ref) + 1e-300
    E = ctx.ell(n, n, K, nnz, cols.ravel(), vals.ravel())
    dx, dy = ctx.vector_from(x), ctx.vector(n)

    def product(M, xv):
        dy.fill(0.0)
        ctx.apply(M, xv, dy)
        ctx.sync()
        return dy.download()

    E.set_kernel(capi.CSR_VECTOR, 1)
    own = product(E, dx)
    ol.assert_parity(own, ref, scale, "ragged ELL, one row per lane")
    base = E.get_param("device_bytes")
    E.set_kernel(capi.CSR_PANEL)  # the row-grouped copy, panel layout forced on it
    assert E.get_param("rowgrouped_kernel") == capi.CSR_PANEL
    held = E.get_param("device_bytes") - base
    assert held < 17 * (nnz + n) + (1 << 20), (held, nnz, n * K)  # (row_ptr + the panel layout of entries + one pad per row; n K = 6M slots)
    ol.assert_parity(product(E, dx), ref, scale, "ragged ELL, row-grouped copy")
    xn = x.copy()
    xn[0] = np.nan
    dxn = ctx.vector_from(xn)
    padded = lens < K
    for kernel, lanes, what in ((capi.CSR_PANEL, 0, "copy"), (capi.CSR_VECTOR, 1, "own kernel")):
        E.set_kernel(kernel, lanes)
        got = product(E, dxn)
        assert np.array_equal(np.isnan(got), padded), what


# ---------------------------------------------------------------------------------- full-size properties

def _abs_row_scale(ctx, pkg, A, x):
    """(|A||x|)_i for every row of a device-resident CSR handle, computed on the device: the values come to the host once,
    go back as |a_ij| in a second handle (row-parallel kernel: no layout, no trial launches), and one product gives the
    scale of the parity gate for ALL rows — what the whole-vector comparisons below divide by."""
    rp, cc, cv = A.download()
    np.abs(cv, out=cv)
    B = ctx.csr(A.info.nrow, A.info.ncol, rp, cc, cv)  # (a shard's row_ptr comes back rebased: the product does not care)
    del rp, cc, cv
    B.set_kernel(pkg.capi.CSR_VECTOR)
    xa = ctx.vector(A.info.ncol)
    ctx.axpby(1.0, x, 0.0, x, xa)  # the tests' x is U(0,1): |x| = x; copied so that the caller's vector is not aliased
    s = ctx.vector(A.info.nrow)
    s.fill(0.0)
    ctx.apply(B, xa, s)
    ctx.sync()
    return s.download()


def _check_boundary_rows(orc, synth, hy, row0, nglob, k, seed, hx, starts, what, halo=2):
    """rows start - halo .. start + halo - 1 around every given row-group start against the oracle (regenerated from the seed)"""
    n = len(hy)
    for g0 in starts:
        lo, hi = max(0, g0 - halo), min(n, g0 + halo)
        if lo >= hi:
            continue
        rp, cc, cv = synth.csr_uniform(row0 + lo, row0 + hi, nglob, k, seed=seed)
        ref, scale = np.zeros(hi - lo), np.zeros(hi - lo)
        ol.csr_spmv(orc, rp, cc, cv, hx, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
        ol.assert_parity(hy[lo:hi], ref, scale, f"{what}: rows {row0 + lo}..{row0 + hi} (a row-group boundary)")

def test_full_size_csr_linearity_and_row_sample(ctx, orc, pkg):
    """BASELINE config 2 shape at reduced N would not exercise int32 limits; run the real N = 10M, 32/row.
    Size-independent checks: (i) a sample of rows recomputed by the oracle from the regenerated rows,
    (ii) linearity A(ax1 + bx2) = aAx1 + bAx2, (iii) accumulation: two calls = 2 * one call."""
    synth = pkg.synth
    n, k = 10_000_000, 32
    A = ctx.gen_csr_uniform(0, n, n, k, band=0, seed=1)
    assert A.info.nnz == n * k
    # the kernel BASELINE's headline is measured with (the model's pick at this size; a change here is a change of the headline)
    assert A.info.kernel == pkg.capi.CSR_PANEL and A.get_param("panel_layout") == 4, (A.info.kernel, A.get_param("panel_layout"))
    perf_expect((A.get_param("panel_unroll"), A.get_param("panel_pipe"), A.get_param("panel_sync")) == (8, 2, 1),
                f"C2: the panel trial kept chunk / order / barrier {A.get_param('panel_unroll')}, {A.get_param('panel_pipe')}, {A.get_param('panel_sync')} (bench records: 8, 2, 1)")
    x1 = ctx.gen_vector(n, seed=1)
    x2 = ctx.gen_vector(n, seed=2)
    y1, y2, y3 = ctx.vector(n), ctx.vector(n), ctx.vector(n)
    for v in (y1, y2, y3):
        v.fill(0.0)
    ctx.apply(A, x1, y1)
    ctx.apply(A, x2, y2)
    x3 = ctx.vector(n)
    ctx.axpby(0.5, x1, -2.0, x2, x3)
    ctx.apply(A, x3, y3)
    ctx.sync()
    h1, h2, h3 = y1.download(), y2.download(), y3.download()
    # (i) rows [r0, r0+2000) at three places, against the oracle on the regenerated rows
    hx1 = synth.vec_uniform(n, seed=1)
    for r0 in (0, 4_999_000, n - 2000):
        rp, cc, cv = synth.csr_uniform(r0, r0 + 2000, n, k, seed=1)
        ref = np.zeros(2000)
        ol.csr_spmv(orc, rp, cc, cv, hx1, ref)
        scale = np.zeros(2000)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx1, scale)
        ol.assert_parity(h1[r0:r0 + 2000], ref, scale, f"full-size rows {r0}..")
    # (ii) linearity; |A||x| <= 32 per row, so 1e-10 * 32 bounds the scaled error generously
    assert np.max(np.abs(h3 - (0.5 * h1 - 2.0 * h2))) <= 1e-10 * 32
    # (iii) y += : a second application doubles y
    ctx.apply(A, x1, y1)
    ctx.sync()
    assert np.max(np.abs(y1.download() - 2.0 * h1)) <= 1e-12 * 32
    # (iv) the rows on both sides of row-group boundaries of the panel layout (where a group's accumulators are written
    # back and the next group's slices begin) against the oracle: every 37th group and the last ones
    assert A.info.kernel == pkg.capi.CSR_PANEL
    groups, rows_per = A.get_param("panel_groups"), A.get_param("panel_rows")
    starts = [g * rows_per for g in list(range(1, groups, 37)) + [groups - 2, groups - 1]]
    _check_boundary_rows(orc, synth, h1, 0, n, k, 1, hx1, starts, "C2 panel")
    # (v) ALL 10M rows: the panel product against the row-parallel kernel (an independent code path: no re-ordered
    # layout, one wavefront slice per row), scaled by (|A||x|)_i computed on the device - a defect confined to one row
    # group, one slice or one run of the layout cannot hide between the oracle's samples
    scale = _abs_row_scale(ctx, pkg, A, x1)
    assert scale.min() > 0.0
    A.set_kernel(pkg.capi.CSR_VECTOR)
    y2.fill(0.0)
    ctx.apply(A, x1, y2)
    ctx.sync()
    hv = y2.download()
    err = np.abs(h1 - hv) / scale
    worst = int(np.argmax(err))
    assert err[worst] <= ol.REL_TOL, f"panel vs row-parallel kernel: row {worst}: {h1[worst]!r} vs {hv[worst]!r}, scaled {err[worst]:.3e}"
    assert np.max(np.abs(h1 - hv)) / np.max(np.abs(hv)) <= ol.REL_TOL


def test_csr_at_the_int32_limit_of_entries(ctx, orc, pkg):
    """67.1M rows x 32 = 2,147,200,000 entries, 283,647 below 2^31: the largest shard the int32 offsets of the
    reference's containers (and of this engine's shards) can hold.  Generated on the device (26 GB), multiplied through
    the automatic choice (panel layout: another 26 GB) and the row-parallel kernel; rows at the start, in the middle
    and at the very end are regenerated on the host and checked against the oracle; the two-phase layout, whose
    padding would pass 2^31, is refused with a message and the handle keeps working"""
    synth, capi = pkg.synth, pkg.capi
    n, k = 67_100_000, 32
    free, _ = ctx.mem_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~60 GB of device memory")
    A = ctx.gen_csr_uniform(0, n, n, k, band=0, seed=5)
    assert A.info.nnz == n * k == 2_147_200_000 and A.info.kernel == capi.CSR_PANEL
    x = ctx.gen_vector(n, seed=5)
    hx = synth.vec_uniform(n, seed=5)
    y = ctx.vector(n)

    def check(what):
        y.fill(0.0)
        ctx.apply(A, x, y)
        ctx.sync()
        for r0 in (0, 33_554_000, n - 1500):
            rp, cc, cv = synth.csr_uniform(r0, r0 + 1500, n, k, seed=5)
            ref, scale = np.zeros(1500), np.zeros(1500)
            ol.csr_spmv(orc, rp, cc, cv, hx, ref)
            ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
            ol.assert_parity(y.download(r0, 1500), ref, scale, f"{what}: rows {r0}..")

    check("panel")
    with pytest.raises(capi.SpmvError, match="too fine|int32|entries"):
        A.set_kernel(capi.CSR_TWOPHASE)
    A.set_kernel(capi.CSR_VECTOR)
    check("row-parallel")


# ---------------------------------------------------------------------------------- panel kernel (no locality)
@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_csr_panel_kernel_matches_reference_golden(ctx, orc, pkg, make):
    c = make()
    g = golden(c["name"])
    rp, cc, cv = _csr_of(orc, c)
    scale = _scale(orc, c)
    # (rows per group, panel width, line sort, layout, unroll, wavefront sync, pipeline order, pace ns)
    for rows, width, srt, aos, unroll, sync, pipe, pace in (
            (0, 0, 1, 0, 8, 2, 0, -1), (0, 0, 0, 0, 4, 0, 1, 0), (37, 16, 1, 0, 2, 1, 0, 0), (1000, 1024, 1, 0, 16, 3, 1, 500),
            (20000, 4096, 0, 0, 4, 1, 1, 0), (5, 48, 1, 0, 8, 2, 0, 0), (3, 16, 1, 0, 2, 2, 1, 0), (500, 512, 1, 0, 8, 0, 1, 2000),
            (64, 64, 1, 0, 2, 0, 1, -1), (0, 0, 1, 0, 8, 0, 0, 300),
            # layout 3: 12-byte packed entries (falls back to three arrays when a slice spans too many columns)
            (0, 0, 1, 3, 8, 0, 1, 0), (37, 16, 1, 3, 4, 0, 0, 0), (5, 48, 1, 3, 16, 0, 1, -1), (1000, 1024, 0, 3, 8, 0, 1, 0),
            (0, 0, 1, 3, 8, 0, 2, 0), (64, 64, 1, 3, 4, 1, 2, -1), (3, 16, 1, 3, 2, 3, 2, 0), (0, 0, 1, 3, 0, -1, -1, -1), (500, 512, 1, 0, 8, 0, 2, 0),
            # the C2 instance (packed, U = 8, gather-first) with every compile-time sync, and through the run-time switch
            (0, 0, 1, 3, 8, 1, 2, 0), (0, 0, 1, 3, 8, 2, 2, 0), (0, 0, 1, 3, 8, 3, 2, 0), (7, 32, 1, 3, 8, 3, 2, 400),
            # layout 4: the packed slices stored in interleaved pairs (8- and 16-byte loads)
            (0, 0, 1, 4, 8, 3, 2, 0), (0, 0, 1, 4, 8, 1, 2, 0), (37, 16, 1, 4, 4, 0, 0, 0), (5, 48, 1, 4, 16, 0, 1, -1), (3, 16, 1, 4, 2, 0, 1, 0),
            (64, 64, 1, 4, 4, 1, 2, -1), (0, 0, 1, 4, 8, 0, 0, 0), (1000, 1024, 1, 4, 8, 0, 1, 0), (0, 0, 1, 4, 0, -1, -1, -1)):
        A = ctx.csr(c["nrow"], c["ncol"], rp, cc, cv)
        A.set_param("panel_rows", rows)
        A.set_param("panel_width", width)
        A.set_param("panel_sort", srt)
        A.set_param("panel_aos", aos)
        A.set_param("panel_unroll", unroll)
        A.set_param("panel_sync", sync)
        A.set_param("panel_pipe", pipe)
        A.set_kernel(pkg.capi.CSR_PANEL)  # (`pace`, the last number of a case, was round 1's clock throttle: deleted in round 5)
        y1, y50 = _apply_n(ctx, A, c["x"], c["nrow"], NUM_TEST)
        what = f"{c['name']} panel rows={rows} width={width} sort={srt} layout={aos} unroll={unroll} sync={sync} pipe={pipe} pace={pace}"
        ol.assert_parity(y1, g["y1_csr"], scale, what + " 1 call")
        ol.assert_parity(y50, g["y50_csr"], scale, what + " 50 calls", reps=NUM_TEST)


def test_csr_panel_packed_layout_large(ctx, orc, pkg):
    """12-byte packed entries on matrices big enough for many slices, pipelined chunks and ragged tails; the sparse
    one (1 entry per row over 8M columns) has slices wider than the packed word and must fall back."""
    capi = pkg.capi
    for n, k, band, expect in ((1_500_000, 24, 0, 3), (1_200_000, 16, 4096, 3), (8_000_000, 1, 0, 0)):
        A = ctx.gen_csr_uniform(0, n, n, k, band=band, seed=5)
        x = ctx.gen_vector(n, seed=6)
        yv, yp = ctx.vector(n), ctx.vector(n)
        yv.fill(0.0)
        yp.fill(0.0)
        A.set_kernel(capi.CSR_VECTOR)
        ctx.apply(A, x, yv)
        for unroll, pipe, layout in ((8, 1, 3), (4, 0, 3), (16, 1, 3), (8, 2, 4), (4, 1, 4), (2, 0, 4)):
            A.set_param("panel_aos", layout)
            A.set_param("panel_unroll", unroll)
            A.set_param("panel_pipe", pipe)
            A.set_kernel(capi.CSR_PANEL)
            # (the wrap-around band has two column clusters in its first and last groups: cut into separate slices)
            assert A.get_param("panel_layout") == (layout if expect else 0), (n, k, band)
            yp.fill(0.0)
            ctx.apply(A, x, yp)
            ctx.sync()
            hv, hp = yv.download(), yp.download()
            # same products, different summation order: bound by |A||x| <= k
            assert np.max(np.abs(hv - hp)) <= ol.REL_TOL * k, (n, k, band, unroll, pipe)
        if expect == 3:
            assert A.get_param("panel_bytes") < 12.2 * n * k


def test_csr_auto_picks_panel_for_large_random_and_agrees_with_vector(ctx, orc, pkg):
    synth, capi = pkg.synth, pkg.capi
    n, k = 2_000_000, 16
    A = ctx.gen_csr_uniform(0, n, n, k, seed=77)
    assert A.info.kernel == capi.CSR_PANEL
    x = ctx.gen_vector(n, seed=77)
    yp, yv = ctx.vector(n), ctx.vector(n)
    yp.fill(0.0)
    yv.fill(0.0)
    ctx.apply(A, x, yp)
    A.set_kernel(capi.CSR_VECTOR)
    ctx.apply(A, x, yv)
    ctx.sync()
    hp, hv = yp.download(), yv.download()
    assert np.max(np.abs(hp - hv)) <= 1e-10 * k  # |A||x| <= k
    hx = synth.vec_uniform(n, seed=77)
    for r0 in (0, 1_234_567, n - 3000):
        rp, cc, cv = synth.csr_uniform(r0, r0 + 3000, n, k, seed=77)
        ref, scale = np.zeros(3000), np.zeros(3000)
        ol.csr_spmv(orc, rp, cc, cv, hx, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
        ol.assert_parity(hp[r0:r0 + 3000], ref, scale, f"panel rows {r0}..")


def test_large_coo_takes_the_panel_path_and_matches_oracle(ctx, orc, pkg):
    """C4-like: power-law rows, x beyond L2 -> AUTO groups the entries by row and uses the panel layout"""
    synth, capi = pkg.synth, pkg.capi
    n = 1_000_000
    A = ctx.gen_coo_powerlaw(n, n, 4096, seed=4)
    assert A.info.sorted_rows == 1
    perf_expect(A.info.kernel == capi.CSR_PANEL, f"C4-like COO: AUTO kept {A.info.kernel} (the row-grouped copy was timed 6x faster in round 5)")
    x = ctx.gen_vector(n, seed=4)
    yp, ys = ctx.vector(n), ctx.vector(n)
    yp.fill(0.0)
    ys.fill(0.0)
    ctx.apply(A, x, yp)
    A.set_kernel(capi.CSR_VECTOR)  # the segmented scan on the same handle: over the copy in column bins (x = 8 MB, beyond an XCD's L2)
    assert A.get_param("coo_column_bins") == 8
    ctx.apply(A, x, ys)
    A.set_param("coo_column_bins", 0)  # and over the handle's own arrays in their order
    yi = ctx.vector(n)
    yi.fill(0.0)
    ctx.apply(A, x, yi)
    ctx.sync()
    hp, hs, hi = yp.download(), ys.download(), yi.download()
    row, col, val = synth.coo_powerlaw(n, n, 4096, seed=4)
    hx = synth.vec_uniform(n, seed=4)
    rp, cc, cv = ol.coo_to_csr(orc, n, row, col, val)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, hx, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
    ol.assert_parity(hp, ref, scale, "large coo, panel path")
    ol.assert_parity(hs, ref, scale, "large coo, segmented scan over column bins")
    ol.assert_parity(hi, ref, scale, "large coo, segmented scan in place")


@pytest.mark.parametrize("nrow,k", [(5000, 64), (1001, 20), (257, 1), (4096, 17)])
def test_dia_tiled_kernel_is_bitwise_oracle_fma(ctx, orc, pkg, nrow, k):
    """the LDS-tiled DIA product adds a row's diagonals left to right from y[i], like the reference"""
    synth = pkg.synth
    A = ctx.gen_dia_banded(nrow, k, seed=6)
    off, _, val = A.download()
    assert np.array_equal(off, np.arange(k) - k // 2)
    assert np.array_equal(val, synth.to_sym(synth._draw(synth.stream_key(6, synth.STREAM_VAL), np.arange(nrow * k, dtype=np.uint64))))
    x = synth.vec_uniform(nrow, seed=6)
    y0 = synth.vec_uniform(nrow, seed=60)
    dy = ctx.vector_from(y0)
    ctx.apply(A, ctx.vector_from(x), dy)
    ctx.sync()
    ref = y0.copy()
    ol.dia_spmv(orc, nrow, ol.i32(off), ol.f64(val), x, ref, fma=True)
    assert np.array_equal(dy.download(), ref)


def test_large_csc_is_regrouped_by_row_and_matches_oracle(ctx, orc, pkg):
    """CSC beyond 2M entries: regrouped by row on the device (panel product); forcing VECTOR keeps the atomic scatter"""
    synth, capi = pkg.synth, pkg.capi
    n, k = 300_000, 16
    rp, col, val = synth.csr_uniform(0, n, n, k, seed=31)
    row = np.repeat(np.arange(n, dtype=np.int32), k)
    cp, cr, cw = ol.coo_to_csc(orc, n, row, col, val)
    x = synth.vec_uniform(n, seed=31)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csc_spmv(orc, cp, cr, cw, x, ref)
    ol.csr_abs_row_sums(orc, rp, col, val, x, scale)
    A = ctx.csc(n, n, cp, cr, cw)
    # 4.8M entries: AUTO times the scatter against the copy grouped by row (round 5; rounds 1-4: the model alone, the panel
    # layout forced on the copy); the copy wins on any box (one atomic on y per entry against a row-grouped product)
    perf_expect(A.info.kernel == capi.CSR_PANEL, f"large CSC: AUTO kept {A.info.kernel}")
    if A.info.kernel == capi.CSR_PANEL:
        assert A.get_param("rowgrouped_kernel") in (1, 2, 3, 4, 5, 6, 7)
        assert A.info.device_bytes > 12 * n * k + 12 * n * k - 1  # CSC arrays + the copy (12-byte entries, or CSR's own 12)
    for kernel, what in ((None, "AUTO"), (capi.CSR_VECTOR, "scatter forced"), (capi.CSR_PANEL, "panel layout forced on the copy"), (capi.CSR_AUTO, "AUTO again")):
        if kernel is not None:
            A.set_kernel(kernel)
        if kernel == capi.CSR_PANEL:
            assert A.get_param("rowgrouped_kernel") == capi.CSR_PANEL
        if kernel == capi.CSR_VECTOR:
            assert A.info.kernel == capi.CSR_VECTOR and A.get_param("rowgrouped_kernel") == 0
        y1, _ = _apply_n(ctx, A, x, n, 1)
        ol.assert_parity(y1, ref, scale, f"large csc, {what}")
    # a small one (the golden cases are below the 64K entries where anything is timed): 100K entries, both paths
    n2, k2 = 12_500, 8
    rp2, col2, val2 = synth.csr_uniform(0, n2, n2, k2, seed=32)
    cp2, cr2, cw2 = ol.coo_to_csc(orc, n2, np.repeat(np.arange(n2, dtype=np.int32), k2), col2, val2)
    x2 = synth.vec_uniform(n2, seed=32)
    ref2, scale2 = np.zeros(n2), np.zeros(n2)
    ol.csc_spmv(orc, cp2, cr2, cw2, x2, ref2)
    ol.csr_abs_row_sums(orc, rp2, col2, val2, x2, scale2)
    S = ctx.csc(n2, n2, cp2, cr2, cw2)
    y1, _ = _apply_n(ctx, S, x2, n2, 1)
    ol.assert_parity(y1, ref2, scale2, f"small csc AUTO (kernel {S.info.kernel}, copy runs {S.get_param('rowgrouped_kernel')})")


def test_malformed_matrices_are_refused_before_any_kernel_indexes_with_them(ctx, pkg):
    """the reference never checks an index; on a GPU an out-of-bounds access can reset the node, so handles are
    validated when they are created (and on demand: spmv_mat_validate)"""
    Err = pkg.capi.SpmvError
    rp = np.array([0, 2, 4], np.int32)
    val = np.ones(4)
    with pytest.raises(Err, match="index out of range"):
        ctx.csr(2, 3, rp, np.array([0, 1, 2, 3], np.int32), val)  # column 3 of 3
    with pytest.raises(Err, match="index out of range"):
        ctx.csr(2, 3, rp, np.array([0, -1, 2, 1], np.int32), val)
    with pytest.raises(Err, match="offsets decrease"):
        ctx.csr(3, 3, np.array([0, 3, 1, 4], np.int32), np.array([0, 1, 2, 0], np.int32), val)
    with pytest.raises(Err, match="index out of range"):
        ctx.coo(2, 3, np.array([0, 2], np.int32), np.array([0, 1], np.int32), np.ones(2))  # row 2 of 2
    with pytest.raises(Err, match="index out of range"):
        ctx.ell(2, 3, 2, 4, np.array([0, 1, 5, 2], np.int32), val)
    with pytest.raises(Err):
        ctx.csc(2, 3, np.array([0, 1, 1, 2], np.int32), np.array([0, 7], np.int32), np.ones(2))
    A = ctx.csr(2, 3, rp, np.array([0, 1, 2, 1], np.int32), val)
    A.validate()  # well-formed


def _nasty_shapes():
    """(name, nrow, ncol, row lengths, column sampler) — shapes that stress the panel layout: one huge row, many empty
    rows, a single column, columns at both ends of a wide range (slices must be cut at the gap), more rows than one
    row group holds with almost nothing in them, duplicates"""
    rng = np.random.RandomState(123)
    out = []
    lens = np.zeros(50, np.int64)
    lens[17] = 300_000
    out.append(("one huge row", 50, 70_000, lens, lambda n: rng.randint(0, 70_000, n)))
    lens = np.zeros(60_000, np.int64)
    lens[rng.choice(60_000, 400, replace=False)] = rng.randint(1, 2000, 400)
    out.append(("mostly empty rows", 60_000, 5_000, lens, lambda n: rng.randint(0, 5_000, n)))
    out.append(("single column", 30_000, 1, rng.randint(0, 4, 30_000), lambda n: np.zeros(n, np.int64)))
    out.append(("two far column clusters", 25_000, 40_000_000, rng.randint(8, 24, 25_000),
                lambda n: np.where(rng.rand(n) < 0.5, rng.randint(0, 3000, n), 40_000_000 - 1 - rng.randint(0, 3000, n))))
    out.append(("sparse wide rows", 45_000, 30_000_000, rng.randint(0, 3, 45_000), lambda n: rng.randint(0, 30_000_000, n)))
    out.append(("duplicates in a few columns", 5_000, 8, rng.randint(0, 200, 5_000), lambda n: rng.randint(0, 8, n)))
    return out


def test_panel_kernel_on_nasty_shapes(ctx, orc, pkg):
    capi = pkg.capi
    for name, nrow, ncol, lens, cols in _nasty_shapes():
        lens = np.asarray(lens, np.int64)
        rp = np.zeros(nrow + 1, np.int32)
        rp[1:] = np.cumsum(lens)
        nnz = int(rp[-1])
        cc = np.asarray(cols(nnz), np.int64).astype(np.int32)
        rng = np.random.RandomState(nnz % 9973)
        cv = rng.uniform(-1, 1, nnz)
        x = rng.uniform(0, 1, ncol) if ncol <= 1_000_000 else None
        if x is None:  # wide x: generate on the device, fetch back for the oracle
            xv = ctx.gen_vector(ncol, seed=5)
            x = xv.download()
        else:
            xv = ctx.vector_from(x)
        ref, scale = np.zeros(nrow), np.zeros(nrow)
        ol.csr_spmv(orc, rp, cc, cv, x, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
        ref2 = ref.copy()
        ol.csr_spmv(orc, rp, cc, cv, x, ref2)
        for layout, unroll, pipe, rows in ((3, 0, -1, 0), (3, 8, 2, 0), (3, 4, 1, 7), (3, 2, 0, 20000), (0, 8, 1, 0), (0, 8, 2, 333), (0, 4, 0, 0),
                                           (4, 8, 2, 0), (4, 4, 1, 7), (4, 2, 0, 20000), (4, 0, -1, 0)):
            A = ctx.csr(nrow, ncol, rp, cc, cv)
            for k, v in (("panel_aos", layout), ("panel_unroll", unroll), ("panel_pipe", pipe), ("panel_rows", rows)):
                A.set_param(k, v)
            A.set_kernel(capi.CSR_PANEL)
            y = ctx.vector(nrow)
            y.fill(0.0)
            ctx.apply(A, xv, y)
            ctx.sync()
            what = f"{name}: layout={layout} (in memory {A.get_param('panel_layout')}) unroll={unroll} pipe={pipe} rows={rows}"
            ol.assert_parity(y.download(), ref, scale, what + " 1 call")
            ctx.apply(A, xv, y)
            ctx.sync()
            ol.assert_parity(y.download(), ref2, scale, what + " 2 calls", reps=2)


def test_packed_layout_pads_do_not_leak_non_finite_x(ctx, orc, pkg):
    """The pads of the 12-byte layout multiply 0.0 by x[slice base]; that product (NaN when x there is inf) goes into
    a spare accumulator that is never written back.  Entries only use columns = 5 mod 16, x is inf on every multiple
    of 16 (all possible slice bases), so any leak would show up as NaN/inf in y."""
    capi = pkg.capi
    rng = np.random.RandomState(77)
    nrow, ncol = 30_000, 20_000_000
    lens = rng.randint(4, 12, nrow)
    rp = np.zeros(nrow + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    near = rng.rand(nnz) < 0.5  # two far clusters: slices are cut at the gap and padded
    cc = (np.where(near, rng.randint(0, 2000, nnz), (ncol // 16 - 1) - rng.randint(0, 2000, nnz)) * 16 + 5).astype(np.int32)
    cv = rng.uniform(-1, 1, nnz)
    x = rng.uniform(0, 1, ncol)
    x[::16] = np.inf
    ref, scale = np.zeros(nrow), np.zeros(nrow)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    assert np.all(np.isfinite(ref))
    xs = x.copy()
    xs[::16] = 0.0
    ol.csr_abs_row_sums(orc, rp, cc, cv, xs, scale)
    A = ctx.csr(nrow, ncol, rp, cc, cv)
    for layout in (3, 4):  # 4: an odd slice count gets an empty partner slice (base 0: x[0] is inf too)
        A.set_param("panel_aos", layout)
        A.set_kernel(capi.CSR_PANEL)
        assert A.get_param("panel_layout") == layout
        assert A.get_param("panel_bytes") > 12 * nnz  # there are pads
        y = ctx.vector(nrow)
        y.fill(0.0)
        ctx.apply(A, ctx.vector_from(x), y)
        ctx.sync()
        got = y.download()
        assert np.all(np.isfinite(got))
        ol.assert_parity(got, ref, scale, f"packed layout {layout}, inf at every slice base")


def test_panel_product_every_order_and_barrier_placement_at_size(ctx, pkg):
    """the chunk pipeline with its order written down (round 5: two register sets in turn, scheduling fences, no pace / guard /
    trace code): every chunk size x order x barrier placement x layout on a matrix with many groups, chunks and ragged tails,
    against the row-parallel kernel; the parameters the removed machinery had are refused by name"""
    capi = pkg.capi
    n, k = 1_000_000, 24
    A = ctx.gen_csr_uniform(0, n, n, k, seed=21)
    x = ctx.gen_vector(n, seed=22)
    yv, yp = ctx.vector(n), ctx.vector(n)
    yv.fill(0.0)
    A.set_kernel(capi.CSR_VECTOR)
    ctx.apply(A, x, yv)
    ctx.sync()
    ref = yv.download()
    for layout in (4, 3, 0):
        A.set_param("panel_aos", layout)
        for unroll in (2, 4, 8, 16):
            for order in (0, 1, 2):
                for sync in (0, 1, 2, 3):
                    A.set_param("panel_unroll", unroll)
                    A.set_param("panel_pipe", order)
                    A.set_param("panel_sync", sync)
                    A.set_kernel(capi.CSR_PANEL)
                    assert A.get_param("panel_layout") == layout and A.get_param("panel_unroll") == min(unroll, 8)
                    assert A.get_param("panel_sync") == (3 if sync == 2 else sync)  # the split barrier of rounds 2-4 runs 3
                    yp.fill(0.0)
                    ctx.apply(A, x, yp)
                    ctx.sync()
                    assert np.max(np.abs(yp.download() - ref)) <= ol.REL_TOL * k, (layout, unroll, order, sync)
    for gone in ("panel_pace_ns", "panel_guard", "panel_stagger", "panel_trace", "panel_legacy"):
        with pytest.raises(capi.SpmvError, match="unknown parameter"):
            A.set_param(gone, 0)


def test_large_ell_with_scattered_columns_runs_the_panel_product(ctx, pkg):
    """ELL whose rows touch columns all over x is gather-bound with one lane per row; such handles are regrouped
    (every slot, padding included) and run the panel kernel.  Banded ELL (C3's shape) keeps the ELL kernel."""
    capi = pkg.capi
    n, k = 1_500_000, 24
    csr = ctx.gen_csr_uniform(0, n, n, k, seed=9)
    E = ctx.csr_to_ell(csr)
    perf_expect(E.info.kernel == capi.CSR_PANEL, f"scattered ELL: AUTO kept {E.info.kernel}")
    E.set_kernel(capi.CSR_PANEL)
    x = ctx.gen_vector(n, seed=10)
    y_csr, y_pan, y_ell = ctx.vector(n), ctx.vector(n), ctx.vector(n)
    for v in (y_csr, y_pan, y_ell):
        v.fill(0.0)
    csr.set_kernel(capi.CSR_VECTOR)
    ctx.apply(csr, x, y_csr)
    ctx.apply(E, x, y_pan)
    E.set_kernel(capi.CSR_VECTOR)  # one lane per row
    assert E.info.kernel == capi.CSR_VECTOR
    ctx.apply(E, x, y_ell)
    ctx.sync()
    ref = y_ell.download()  # the ELL kernel adds in the reference's order (bit-identical to the fma oracle, tested above)
    assert np.max(np.abs(y_pan.download() - ref)) <= ol.REL_TOL * k
    assert np.max(np.abs(y_csr.download() - ref)) <= ol.REL_TOL * k
    E.set_kernel(capi.CSR_AUTO)
    perf_expect(E.info.kernel == capi.CSR_PANEL, f"scattered ELL: AUTO again kept {E.info.kernel}")
    B = ctx.gen_ell_banded(1_000_000, 1_000_000, 16, seed=2)
    assert B.info.kernel == capi.CSR_VECTOR
    # ragged rows: padding slots (col 0, val 0.0) are kept, so x[0] = inf poisons every padded row as in the reference
    rp = np.zeros(4001, np.int32)
    lens = np.random.RandomState(3).randint(0, 600, 4000)
    rp[1:] = np.cumsum(lens)
    cc = np.random.RandomState(4).randint(1, 3_000_000, rp[-1]).astype(np.int32)
    cv = np.random.RandomState(5).uniform(0.5, 1.0, rp[-1])
    R = ctx.csr_to_ell(ctx.csr(4000, 3_000_000, rp, cc, cv))
    R.set_kernel(capi.CSR_PANEL)
    xs = np.ones(3_000_000)
    xs[0] = np.inf
    yp, ye = ctx.vector(4000), ctx.vector(4000)
    yp.fill(0.0)
    ye.fill(0.0)
    ctx.apply(R, ctx.vector_from(xs), yp)
    R.set_kernel(capi.CSR_VECTOR)
    ctx.apply(R, ctx.vector_from(xs), ye)
    ctx.sync()
    a, b = yp.download(), ye.download()
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).sum() > 0
    ok = ~np.isnan(a)
    assert np.max(np.abs(a[ok] - b[ok])) <= ol.REL_TOL * 600


def test_csr_split_columns_is_exact_and_the_parts_add_up(ctx, orc, pkg):
    """spmv_csr_split_columns (sharded solver step): arrays equal a numpy split (order inside rows kept, inside part
    rebased), and inside * x[c0:c1] + outside * x equals the unsplit product within the parity gate"""
    synth = pkg.synth
    n, k = 120_000, 12
    rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=14)
    x = synth.vec_uniform(n, seed=14)
    A = ctx.csr(n, n, rp, cc, cv)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    for c0, c1 in ((30_000, 75_000), (0, n), (0, 0), (n - 7, n)):
        A_in, A_out = ctx.csr_split_columns(A, c0, c1)
        inside = (cc >= c0) & (cc < c1)
        rows = np.repeat(np.arange(n), np.diff(rp))
        for part, mask, rebase in ((A_in, inside, c0), (A_out, ~inside, 0)):
            prp, pcc, pcv = part.download()
            erp = np.zeros(n + 1, np.int64)
            np.add.at(erp, rows[mask] + 1, 1)
            assert np.array_equal(prp, np.cumsum(erp)) and np.array_equal(pcc, cc[mask] - rebase) and np.array_equal(pcv, cv[mask])
        assert A_in.info.ncol == c1 - c0 and A_out.info.ncol == n
        y = ctx.vector(n)
        y.fill(0.0)
        ctx.apply(A_in, ctx.vector_from(x[c0:c1]), y)
        ctx.apply(A_out, ctx.vector_from(x), y)
        ctx.sync()
        ol.assert_parity(y.download(), ref, scale, f"split [{c0},{c1})")
    with pytest.raises(pkg.capi.SpmvError):
        ctx.csr_split_columns(A, 5, n + 1)
    # a row shard split at a column range that does not start at the shard's first row: `inside` reports
    # row_begin - col_begin (the column at which local row 0 would meet its diagonal), `outside` the shard's own row_begin
    b, e = 40_000, 70_000
    S = ctx.csr_shard(b, e, n, rp.astype(np.int64), cc, cv)
    for c0, c1, want in ((b, e, 0), (10_000, 90_000, 30_000), (55_000, 99_000, -15_000)):
        s_in, s_out = ctx.csr_split_columns(S, c0, c1)
        assert s_in.info.row_begin == want == b - c0 and s_out.info.row_begin == b
        assert s_in.info.nrow == s_out.info.nrow == e - b and s_in.info.ncol == c1 - c0
        y = ctx.vector(e - b)
        y.fill(0.0)
        ctx.apply(s_in, ctx.vector_from(x[c0:c1]), y)
        ctx.apply(s_out, ctx.vector_from(x), y)
        ctx.sync()
        ol.assert_parity(y.download(), ref[b:e], scale[b:e], f"shard [{b},{e}) split at [{c0},{c1})")


def test_twophase_stream_in_four_pieces_agrees_with_the_row_parallel_kernel(ctx, orc, pkg):
    """The largest product stream the two-phase layout holds takes all four 1 GB pieces of its table (just under 2^29 padded
    entries): 12.6M rows x 32 over 100M columns.  Every piece boundary lies inside some row group's stretch; the product is
    compared with the row-parallel kernel over ALL rows (scaled by (|A||x|)_i) and with the oracle on sampled rows."""
    synth, capi = pkg.synth, pkg.capi
    n, ncol, k = 12_600_000, 100_000_000, 32
    free, _ = ctx.mem_info()
    if free < 60 * 2**30:
        pytest.skip("needs ~35 GB of device memory")
    A = ctx.gen_csr_uniform(0, n, ncol, k, seed=11)
    assert A.info.kernel == capi.CSR_TWOPHASE and A.get_param("twophase_pieces") == 4
    assert 3 * 2**27 < A.get_param("twophase_padded") < 2**29
    x = ctx.gen_vector(ncol, seed=11)
    hx = synth.vec_uniform(ncol, seed=11)
    y, yv = ctx.vector(n), ctx.vector(n)
    y.fill(0.0)
    ctx.apply(A, x, y)
    ctx.sync()
    hy = y.download()
    for r0 in (0, 6_300_000, n - 1000):
        rp, cc, cv = synth.csr_uniform(r0, r0 + 1000, ncol, k, seed=11)
        ref, scale = np.zeros(1000), np.zeros(1000)
        ol.csr_spmv(orc, rp, cc, cv, hx, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
        ol.assert_parity(hy[r0:r0 + 1000], ref, scale, f"four pieces: rows {r0}..")
    scale = _abs_row_scale(ctx, pkg, A, x)
    A.set_kernel(capi.CSR_VECTOR)
    yv.fill(0.0)
    ctx.apply(A, x, yv)
    ctx.sync()
    hv = yv.download()
    err = np.abs(hy - hv) / scale
    worst = int(np.argmax(err))
    assert err[worst] <= ol.REL_TOL, f"two-phase (4 pieces) vs row-parallel: row {worst}: {hy[worst]!r} vs {hv[worst]!r}, scaled {err[worst]:.3e}"


def test_twophase_piece_search_is_bounded_and_gives_its_memory_back(ctx, orc, pkg, monkeypatch):
    """The two-phase layout's product stream lives in 1 GB pieces chosen by timing configurations of a pool: the
    pool is transient - after the build the device holds the handle's own bytes and nothing of the budget - the outcome is
    on record, a budget of 0 means no timing launches at all, and whichever pieces are kept the product is the same."""
    import gc

    capi, synth = pkg.capi, pkg.synth
    for name in ("SPMV_TP_PLACEMENT_BUDGET_MB", "SPMV_PANEL_TRIAL"):  # (the env sweeps of tools/env_sweeps.sh set these: this test is about the defaults)
        monkeypatch.delenv(name, raising=False)
    n, ncol, k = 2_500_000, 40_000_000, 32  # 80M entries: a stream of 0.66 GB (one piece), searched because it is >= 512 MB
    gc.collect()
    ctx.sync()
    free0, _ = ctx.mem_info()
    A = ctx.gen_csr_uniform(0, n, ncol, k, seed=9)
    ctx.sync()
    free1, _ = ctx.mem_info()
    assert A.info.kernel == capi.CSR_TWOPHASE and A.get_param("twophase_pieces") == 1
    held = A.get_param("device_bytes")
    # nothing of the 8 GB budget (nor the scratch vectors) is still held: the device's free memory fell by the handle's bytes
    assert abs((free0 - free1) - held) < 300 << 20, ((free0 - free1) >> 20, held >> 20)
    assert A.get_param("twophase_placements_timed") >= 2 and A.get_param("twophase_placement_spread") >= 1000  # (the kept configuration held up against the one as built, or the latter stayed)
    assert A.get_param("twophase_placement_budget_mb") == -1  # the default: 8192 MB unless the environment says otherwise
    x, y, yv = ctx.gen_vector(ncol, seed=9), ctx.vector(n), ctx.vector(n)
    hx = synth.vec_uniform(ncol, seed=9)

    def check(what):
        y.fill(0.0)
        ctx.apply(A, x, y)
        ctx.sync()
        hy = y.download()
        for r0 in (0, 1_234_000, n - 1500):
            rp, cc, cv = synth.csr_uniform(r0, r0 + 1500, ncol, k, seed=9)
            ref, scale = np.zeros(1500), np.zeros(1500)
            ol.csr_spmv(orc, rp, cc, cv, hx, ref)
            ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
            ol.assert_parity(hy[r0:r0 + 1500], ref, scale, f"{what}: rows {r0}..")
        return hy

    first = check("pieces as chosen at build")
    # the search again with a smaller budget, and switched off: same product, the record follows
    A.set_param("twophase_placement_budget_mb", 3072)
    A.set_param("twophase_choose_pieces", 1)
    assert A.get_param("twophase_placements_timed") >= 2
    again = check("pieces chosen again within 3 GB")
    A.set_param("twophase_placement_budget_mb", 0)
    A.set_param("twophase_choose_pieces", 1)
    assert A.get_param("twophase_placements_timed") == 0
    off = check("no search")
    scale_all = np.maximum(np.abs(first), 1e-300)
    assert np.max(np.abs(first - again) / (scale_all + 32.0)) <= ol.REL_TOL and np.max(np.abs(first - off) / (scale_all + 32.0)) <= ol.REL_TOL
    ctx.sync()
    free2, _ = ctx.mem_info()
    assert abs((free0 - free2) - A.get_param("device_bytes")) < 1500 << 20  # (x, y, yv of this test: 0.36 GB)
    monkeypatch.delenv("SPMV_EXPERIMENTS", raising=False)
    with pytest.raises(capi.SpmvError, match="experiment"):
        A.set_param("twophase_only", 1)  # refused without SPMV_EXPERIMENTS=1: it would make the product wrong
    del A, x, y, yv


@pytest.mark.parametrize("offer", [0, 1, 2], ids=["released", "offered", "carved piece forced under the stream"])
def test_twophase_stream_may_move_into_the_csr_copy_it_releases(ctx, orc, pkg, monkeypatch, offer):
    """panel_keep_csr = 0 on a two-phase handle (round 5): the whole gigabytes inside col_ind / values are candidates for the
    product stream before they are released - memory from another moment of the allocator's history at no transient cost.
    Whatever the search decides (0: not offered; 1: offered, the timings decide; 2: a carved piece forced under slot 0), the
    product is the oracle's, the handle's byte count equals what the device actually lost, and everything comes back with it."""
    import gc

    capi, synth = pkg.capi, pkg.synth
    for name in ("SPMV_TP_PLACEMENT_BUDGET_MB", "SPMV_PANEL_TRIAL"):
        monkeypatch.delenv(name, raising=False)
    n, ncol, k = 4_500_000, 40_000_000, 32  # 144M entries: values 1.15 GB (one whole gigabyte to carve), a stream of two pieces
    gc.collect()
    ctx.sync()
    free0, _ = ctx.mem_info()
    A = ctx.gen_csr_uniform(0, n, ncol, k, seed=19)
    assert A.info.kernel == capi.CSR_TWOPHASE and A.get_param("twophase_pieces") == 2 and A.get_param("twophase_pieces_carved") == 0
    timed_at_build = A.get_param("twophase_placements_timed")
    A.set_param("twophase_offer_csr_copy", offer)
    A.set_param("panel_keep_csr", 0)
    ctx.sync()
    carved = A.get_param("twophase_pieces_carved")
    assert carved == 0 if offer == 0 else (carved >= 1 if offer == 2 else carved in (0, 1))
    if offer:
        assert A.get_param("twophase_placements_timed") > timed_at_build and A.get_param("twophase_placement_spread") >= 1000
    free1, _ = ctx.mem_info()
    held = A.get_param("device_bytes")
    assert abs((free0 - free1) - held) < 300 << 20, ((free0 - free1) >> 20, held >> 20, carved)
    # a carved piece keeps its whole parent (values: 1.15 GB for 1 GB of stream) and gives one fresh gigabyte back
    matrix = 12 * n * k
    assert held <= 1.0 * matrix + 2.2e9 + (0.2e9 if carved else 0) + 8 * n * 2, (held, matrix)
    x, y = ctx.gen_vector(ncol, seed=19), ctx.vector(n)
    hx = synth.vec_uniform(ncol, seed=19)
    for rep in range(2):
        y.fill(0.0)
        ctx.apply(A, x, y)
        ctx.sync()
        hy = y.download()
        for r0 in (0, 2_222_000, n - 1500):
            rp, cc, cv = synth.csr_uniform(r0, r0 + 1500, ncol, k, seed=19)
            ref, scale = np.zeros(1500), np.zeros(1500)
            ol.csr_spmv(orc, rp, cc, cv, hx, ref)
            ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
            ol.assert_parity(hy[r0:r0 + 1500], ref, scale, f"offer {offer}, carved {carved}, product {rep}: rows {r0}..")
    with pytest.raises(capi.SpmvError):
        A.download()  # the CSR arrays are gone either way (released, or under the stream)
    del A, x, y
    gc.collect()
    ctx.sync()
    free2, _ = ctx.mem_info()
    assert abs(free0 - free2) < 300 << 20, ((free0 - free2) >> 20,)


def test_handles_give_their_device_memory_back(ctx, pkg):
    """every layout a handle builds (CSR arrays, panel copy, packed words, slice tables, guard words, the regrouped
    copies of COO / ELL handles, solver work vectors) is released with it"""
    import gc

    capi = pkg.capi
    gc.collect()
    ctx.sync()
    free0, total = ctx.mem_info()
    assert 0 < free0 <= total
    for _ in range(3):
        A = ctx.gen_csr_uniform(0, 1_500_000, 1_500_000, 16, seed=4)  # panel layout + trials
        for layout in (0, 3, 4):
            A.set_param("panel_aos", layout)
            A.set_kernel(capi.CSR_PANEL)
        E = ctx.csr_to_ell(A)  # scattered columns: regrouped copy
        P = ctx.gen_coo_powerlaw(300_000, 300_000, 2048, seed=2)  # COO with its row-grouped copy
        a_in, a_out = ctx.csr_split_columns(A, 100, 700_000)
        x, b = ctx.vector(1_500_000), ctx.gen_vector(1_500_000, seed=1)
        x.fill(0.0)
        try:
            ctx.cg(A, b, x, max_iter=3, rel_tol=0.0)  # not SPD: may stop with an error after allocating its vectors
        except capi.SpmvError:
            pass
        del A, E, P, a_in, a_out, x, b
        gc.collect()
    ctx.sync()
    free1, _ = ctx.mem_info()
    assert free0 - free1 < 256 << 20, f"{(free0 - free1) >> 20} MiB of device memory not returned"


def test_full_size_c3_ell_and_c4_coo_row_samples(ctx, orc, pkg):
    """BASELINE configs 3 and 4 at full size (ELL N = 4M, K = 64 circulant band; COO N = 2M power-law rows, max 4096):
    rows sampled across the matrix are regenerated on the host from the seed (counter-based generators) and checked
    against the oracle's product; plus linearity A(2x) = 2 Ax over the whole vector"""
    synth = pkg.synth
    # ---- C3
    n, k = 4_000_000, 64
    E = ctx.gen_ell_banded(n, n, k, seed=1)
    # C3 runs the format's own kernel over slots recognised as diagonals (which variant of it - two rows per lane, or the DIA-order
    # copy - is the trial's to say: both add in slot order)
    assert E.info.kernel == pkg.capi.CSR_VECTOR and E.get_param("ell_diagonal_slots") == 1 and E.get_param("ell_variant") in (0, 1, 2, 3)
    perf_expect(E.get_param("ell_variant") in (0, 3), f"C3: the trial kept variant {E.get_param('ell_variant')}")
    x = ctx.gen_vector(n, seed=1)
    hx = synth.vec_uniform(n, seed=1)
    y = ctx.vector(n)
    y.fill(0.0)
    ctx.apply(E, x, y)
    ctx.sync()
    hy = y.download()
    key = synth.stream_key(1, synth.STREAM_VAL)
    for r0 in (0, 1_234_567, n - 3000):
        rows = np.arange(r0, r0 + 3000, dtype=np.int64)
        d = np.arange(k, dtype=np.int64)
        cc = ((rows[:, None] + d[None, :] - k // 2) % n).astype(np.int32).ravel()
        cv = synth.to_sym(synth._draw(key, (rows[:, None] * k + d[None, :]).astype(np.uint64).ravel()))
        rp = (np.arange(len(rows) + 1) * k).astype(np.int32)
        ref, scale = np.zeros(len(rows)), np.zeros(len(rows))
        ol.csr_spmv(orc, rp, cc, cv, hx, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
        ol.assert_parity(hy[r0:r0 + 3000], ref, scale, f"C3 rows {r0}..")
    # the format's own variants and the DIA-order copy add a row's products in the same (slot) order: the whole vector, bit for bit
    variant = E.get_param("ell_variant")
    E.set_param("ell_dia_order", 1)
    assert E.get_param("ell_dia_order") == 1 and E.get_param("ell_non_conforming_rows") == k - 1  # the wrap-around rows of the circulant band: 32 at the top, 31 at the bottom
    yd = ctx.vector(n)
    yd.fill(0.0)
    ctx.apply(E, x, yd)
    E.set_kernel(1, 2)  # two rows per lane over the column-major values (a copy that was ASKED for stays allocated; one the trial kept would go back)
    yv = ctx.vector(n)
    yv.fill(0.0)
    ctx.apply(E, x, yv)
    ctx.sync()
    assert np.array_equal(yd.download(), hy) and np.array_equal(yv.download(), hy), f"C3: AUTO ran variant {variant}"
    del yd, yv
    x2, y2 = ctx.vector(n), ctx.vector(n)
    ctx.axpby(2.0, x, 0.0, x, x2)
    y2.fill(0.0)
    ctx.apply(E, x2, y2)
    ctx.sync()
    assert np.max(np.abs(y2.download() - 2.0 * hy)) <= ol.REL_TOL * k
    del E, x, y, x2, y2
    # ---- C4
    n, max_len = 2_000_000, 4096
    P = ctx.gen_coo_powerlaw(n, n, max_len, seed=1)
    lens = synth.powerlaw_lengths(n, max_len, 1).astype(np.int64)
    assert P.info.nnz == int(lens.sum()) and P.info.sorted_rows == 1
    perf_expect(P.info.kernel == pkg.capi.CSR_PANEL and P.get_param("rowgrouped_kernel") == pkg.capi.CSR_PANEL,
                f"C4: AUTO kept kernel {P.info.kernel}, the copy runs {P.get_param('rowgrouped_kernel')} (bench records: the panel layout on the row-grouped copy)")
    x = ctx.gen_vector(n, seed=1)
    hx = synth.vec_uniform(n, seed=1)
    y = ctx.vector(n)
    y.fill(0.0)
    ctx.apply(P, x, y)
    ctx.sync()
    hy = y.download()
    kc, kv = synth.stream_key(1, synth.STREAM_COL), synth.stream_key(1, synth.STREAM_VAL)
    for r0 in (0, 777_777, n - 2500):
        rows = np.arange(r0, r0 + 2500, dtype=np.int64)
        ln = lens[r0:r0 + 2500]
        rr = np.repeat(rows, ln)
        start = np.concatenate(([0], np.cumsum(ln)))
        s = np.arange(rr.size, dtype=np.int64) - np.repeat(start[:-1], ln)
        gidx = (rr * max_len + s).astype(np.uint64)
        cc = synth.to_range(synth._draw(kc, gidx), n).astype(np.int32)
        cv = synth.to_sym(synth._draw(kv, gidx))
        rp = start.astype(np.int32)
        ref, scale = np.zeros(len(rows)), np.zeros(len(rows))
        ol.csr_spmv(orc, rp, cc, cv, hx, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
        ol.assert_parity(hy[r0:r0 + 2500], ref, scale, f"C4 rows {r0}..")


def test_full_size_c5_last_shard_row_sample(ctx, orc, pkg):
    """BASELINE config 5, the shard of the LAST of 8 ranks at full size: rows [70M, 80M) of the 80M x 80M matrix (global
    row / entry indices beyond 2^31, x of 640 MB).  Sampled rows against the oracle on rows regenerated from the seed."""
    synth = pkg.synth
    nglob, k = 80_000_000, 32
    b, e = 70_000_000, 80_000_000
    A = ctx.gen_csr_uniform(b, e, nglob, k, seed=1)
    # x (640 MB) is 8x what the shard's rows can keep busy per sweep: the automatic choice is the two-phase kernel
    assert A.info.nnz == (e - b) * k and A.info.row_begin == b and A.info.kernel == pkg.capi.CSR_TWOPHASE
    x = ctx.gen_vector(nglob, seed=1)
    hx = synth.vec_uniform(nglob, seed=1)
    y = ctx.vector(e - b)
    got = {}
    for kernel in (pkg.capi.CSR_TWOPHASE, pkg.capi.CSR_PANEL):  # both at full size
        A.set_kernel(kernel)
        y.fill(0.0)
        ctx.apply(A, x, y)
        ctx.sync()
        hy = y.download()
        got[kernel] = hy
        for r0 in (b, b + 4_321_000, e - 2000):
            rp, cc, cv = synth.csr_uniform(r0, r0 + 2000, nglob, k, seed=1)
            ref, scale = np.zeros(2000), np.zeros(2000)
            ol.csr_spmv(orc, rp, cc, cv, hx, ref)
            ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
            ol.assert_parity(hy[r0 - b:r0 - b + 2000], ref, scale, f"C5 shard rows {r0}.. kernel {kernel}")
        # rows on both sides of row-group boundaries (two-phase: equal groups of ceil(10M / 512) rows, a group's stretch of
        # the product stream ends and the next begins; panel: the groups of its own layout) against the oracle
        if kernel == pkg.capi.CSR_PANEL:
            groups, per = A.get_param("panel_groups"), A.get_param("panel_rows")
        else:
            per = -(-(e - b) // 512)
            groups = -(-(e - b) // per)
        starts = [g * per for g in list(range(1, groups, 41)) + [groups - 2, groups - 1]]
        _check_boundary_rows(orc, synth, hy, b, nglob, k, 1, hx, starts, f"C5 shard kernel {kernel}")
    # ALL 10M rows: the two products come from layouts that share nothing (product stream in (group, panel) runs against
    # line-sorted packed slices) - compared row by row, scaled by (|A||x|)_i computed on the device
    scale = _abs_row_scale(ctx, pkg, A, x)
    assert scale.min() > 0.0
    ya, yb = got[pkg.capi.CSR_TWOPHASE], got[pkg.capi.CSR_PANEL]
    err = np.abs(ya - yb) / scale
    worst = int(np.argmax(err))
    assert err[worst] <= ol.REL_TOL, f"two-phase vs panel: row {b + worst}: {ya[worst]!r} vs {yb[worst]!r}, scaled {err[worst]:.3e}"
    assert np.max(np.abs(ya - yb)) / np.max(np.abs(yb)) <= ol.REL_TOL
