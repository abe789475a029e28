"""Pin the oracle: our C restatement (oracle/spmv_oracle.c) against the golden vectors produced by the REAL
reference (tests/golden/*.npz, made by tests/golden/make_golden.py from oracle/_ref) — bit for bit — and,
where the compiled reference is present (build container; the .so also travels to the GPU box), live against
it on fresh random inputs.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle_lib as ol
from conftest import golden

NUM_TEST = 50


def _inputs(c):
    return c["nrow"], c["ncol"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]), ol.f64(c["x"])


def _acc(fn, nrow):
    y = np.zeros(nrow)
    fn(y)
    y1 = y.copy()
    for _ in range(NUM_TEST - 1):
        fn(y)
    return y1, y


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_inputs_match_fixture(make):
    c = make()
    g = golden(c["name"])
    if c["name"] == "c1":
        assert cases.digest(ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]), ol.f64(c["x"])) == str(g["sha_inputs"])
    else:
        for key in ("row", "col", "val", "x"):
            assert np.array_equal(c[key], g[key]), key


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_conversions_bitwise_equal_reference(orc, make):
    c = make()
    g = golden(c["name"])
    nrow, ncol, row, col, val, _ = _inputs(c)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    cp, cr, cw = ol.coo_to_csc(orc, ncol, row, col, val)
    k, ec, ev = ol.coo_to_ell(orc, nrow, row, col, val)
    assert k == int(g["ell_k"])
    if c["name"] == "c1":
        assert cases.digest(rp, cc, cv) == str(g["sha_csr"])
        assert cases.digest(cp, cr, cw) == str(g["sha_csc"])
        assert cases.digest(ec, ev) == str(g["sha_ell"])
    else:
        assert np.array_equal(rp, g["csr_row_ptr"]) and np.array_equal(cc, g["csr_col"]) and np.array_equal(cv, g["csr_val"])
        assert np.array_equal(cp, g["csc_col_ptr"]) and np.array_equal(cr, g["csc_row"]) and np.array_equal(cw, g["csc_val"])
        assert np.array_equal(ec, g["ell_col"]) and np.array_equal(ev, g["ell_val"])


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_spmv_bitwise_equal_reference_after_1_and_50_calls(orc, make):
    c = make()
    g = golden(c["name"])
    nrow, ncol, row, col, val, x = _inputs(c)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    cp, cr, cw = ol.coo_to_csc(orc, ncol, row, col, val)
    k, ec, ev = ol.coo_to_ell(orc, nrow, row, col, val)
    for fmt, fn in (("coo", lambda y: ol.coo_spmv(orc, row, col, val, x, y)),
                    ("csr", lambda y: ol.csr_spmv(orc, rp, cc, cv, x, y)),
                    ("csc", lambda y: ol.csc_spmv(orc, cp, cr, cw, x, y)),
                    ("ell", lambda y: ol.ell_spmv(orc, nrow, k, ec, ev, x, y))):
        y1, y50 = _acc(fn, nrow)
        assert np.array_equal(y1, g[f"y1_{fmt}"]), f"{c['name']} {fmt} after 1 call"
        assert np.array_equal(y50, g[f"y50_{fmt}"]), f"{c['name']} {fmt} after 50 calls"


def test_fma_flavour_stays_within_the_parity_tolerance(orc):
    """the _fma flavour (aarch64 contraction; what the HIP kernels compute) vs the pinned plain flavour"""
    for make in cases.ALL_CASES:
        c = make()
        nrow, ncol, row, col, val, x = _inputs(c)
        rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
        scale = np.zeros(nrow)
        ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
        a, b = np.zeros(nrow), np.zeros(nrow)
        ol.csr_spmv(orc, rp, cc, cv, x, a)
        ol.csr_spmv(orc, rp, cc, cv, x, b, fma=True)
        ol.assert_parity(b, a, scale, c["name"] + " fma vs plain")


def test_dia_and_blas1_against_reference_golden(orc):
    c = cases.tri8()
    g = golden("tri8")
    nrow, ncol, row, col, val, x = _inputs(c)
    rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
    off, dv = ol.csr_to_dia(orc, nrow, ncol, rp, cc, cv)
    assert np.array_equal(off, g["dia_offsets"]) and np.array_equal(dv, g["dia_val"])
    y1, y50 = _acc(lambda y: ol.dia_spmv(orc, nrow, off, dv, x, y), nrow)
    assert np.array_equal(y1, g["y1_dia"]) and np.array_equal(y50, g["y50_dia"])
    assert ol.dot(orc, x, x) == float(g["dot_xx"])
    yv = ol.f64(g["y1_csr"])
    for tag, (a, b) in dict(g=(0.75, -1.25), a0=(0.0, 2.0), b0=(3.0, 0.0), a1=(1.0, 0.5), am1=(-1.0, 0.5), b1=(0.5, 1.0),
                            bm1=(0.5, -1.0)).items():
        w = np.zeros(nrow)
        ol.axpby(orc, a, x, b, yv, w)
        assert np.array_equal(w, g[f"axpby_{tag}"]), tag


def test_partition_is_the_numa_drivers(orc):
    # src/mat_vec.cpp:233,245-246: nrow / nthreads rows each, the last takes the remainder
    assert [ol.partition_rows(orc, 10, 4, p) for p in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]
    assert [ol.partition_rows(orc, 3, 8, p) for p in range(8)] == [(0, 0)] * 7 + [(0, 3)]
    rp = np.array([0, 2, 2, 5, 9, 9, 12], np.int32)
    assert np.array_equal(ol.csr_shard_row_ptr(orc, rp, 2, 5), [0, 3, 7, 7])


@pytest.mark.skipif(not ol.ref_available(), reason="oracle/_ref not built (needs the reference sources)")
def test_oracle_live_against_compiled_reference():
    """fresh random matrices, oracle vs the reference's own compiled code, bit for bit"""
    orc, ref = ol.load_oracle(), ol.load_ref()
    p = ol._p
    rng = np.random.RandomState(123)
    for nrow, ncol, nnz in ((1, 1, 1), (17, 5, 60), (200, 300, 5000), (1000, 1000, 16000)):
        row = rng.randint(0, nrow, size=nnz).astype(np.int32)
        col = rng.randint(0, ncol, size=nnz).astype(np.int32)
        val = rng.uniform(-1, 1, size=nnz)
        x = rng.uniform(0, 1, size=ncol)
        rp, cc, cv = ol.coo_to_csr(orc, nrow, row, col, val)
        rrp, rcc, rcv = np.zeros(nrow + 1, np.int32), np.zeros(nnz, np.int32), np.zeros(nnz)
        ref.ref_coo_to_csr(nrow, ncol, nnz, p(row), p(col), p(val), p(rrp), p(rcc), p(rcv))
        assert np.array_equal(rp, rrp) and np.array_equal(cc, rcc) and np.array_equal(cv, rcv)
        k, ec, ev = ol.coo_to_ell(orc, nrow, row, col, val)
        assert k == ref.ref_coo_to_ell(nrow, ncol, nnz, p(row), p(col), p(val), None, None)
        a, b = np.zeros(nrow), np.zeros(nrow)
        ol.coo_spmv(orc, row, col, val, x, a)
        ref.ref_coo_spmv(nrow, ncol, nnz, p(row), p(col), p(val), p(x), p(b))
        assert np.array_equal(a, b)
        a[:] = 0
        b[:] = 0
        ol.csr_spmv(orc, rp, cc, cv, x, a)
        ref.ref_csr_spmv(nrow, ncol, p(rp), p(cc), p(cv), p(x), p(b))
        assert np.array_equal(a, b)
        a[:] = 0
        b[:] = 0
        ol.ell_spmv(orc, nrow, k, ec, ev, x, a)
        ref.ref_ell_spmv(nrow, ncol, nnz, k, p(ec), p(ev), p(x), p(b))
        assert np.array_equal(a, b)
        n = min(nrow, ncol)
        assert ol.dot(orc, x[:n], x[:n]) == ref.ref_dot(n, p(x[:n].copy()), p(x[:n].copy()))


def test_first_touch_copy_of_the_cpu_baseline_is_a_faithful_copy(orc):
    """orc_csr_first_touch_copy (bench.py's cpu_baseline leg: arrays placed by the OpenMP team that multiplies them) must
    hand the reference's loop exactly the matrix and x it was given, and a zeroed y"""
    import ctypes as C

    rng = np.random.default_rng(12)
    nrow, ncol = 5003, 7001
    lens = rng.integers(0, 9, nrow)
    rp = np.zeros(nrow + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    nnz = int(rp[-1])
    col = rng.integers(0, ncol, nnz).astype(np.int32)
    val, x = rng.uniform(-1, 1, nnz), rng.uniform(0, 1, ncol)
    d_rp, d_col, d_val = np.full(nrow + 1, -7, np.int32), np.full(nnz, -7, np.int32), np.full(nnz, np.nan)
    d_x, d_y = np.full(ncol, np.nan), np.full(nrow, np.nan)
    orc.orc_csr_first_touch_copy.restype = None
    p = ol._p
    orc.orc_csr_first_touch_copy(C.c_int32(nrow), C.c_int64(ncol), p(rp), p(col), p(val), p(x), p(d_rp), p(d_col), p(d_val), p(d_x), p(d_y))
    assert np.array_equal(d_rp, rp) and np.array_equal(d_col, col) and np.array_equal(d_val, val) and np.array_equal(d_x, x)
    assert not d_y.any()
