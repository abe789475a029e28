"""spmv_apply_host - the reference's own call shape (include/mat_vec.h:7-11: host vectors on every call; main.cpp:56-59
does it 50 times) - over EVERY kernel a handle may end up running, the ones that add into y with device atomics included.

Round 5's review found the staged path (x + y <= 1 MB) deciding where y lives from the handle's FORMAT: a CSR handle under
SEGSCAN or SPLIT's chunks, or a COO / CSC / ELL handle whose row-grouped copy had picked one of those, ran
global_atomic_add_f64 on pinned HOST memory - platform behaviour, not a HIP guarantee.  The engine now decides from the kernel
that runs ("adds_into_y_with_atomics", followed through a handle's copies) and stages y in device memory there.  These tests
run hub-row and arrow matrices through apply_host under every forced kernel and as COO / CSC / ELL handles, 1 and 50 calls,
against the oracle within the parity gate, with CPU stores of x into device memory on and off (SPMV_HOST_STORES)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
NUM_TEST = 50  # main.cpp:16


def _hub(rng):
    """40000 rows x 8 entries and one row of 30000 (test_gpu_parity.py's hub matrix): x + y = 640 KB, the staged path"""
    n = 40_000
    lens = np.full(n, 8, np.int64)
    lens[1234] = 30_000
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cc = rng.integers(0, n, rp[-1]).astype(np.int32)
    cv = rng.uniform(-1, 1, rp[-1])
    return n, lens, rp, cc, cv


def _arrow(rng):
    """dense first row, dense first column, a diagonal, 2000 empty rows; 60000 rows: x + y = 960 KB, the staged path"""
    n = 60_000
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
    return n, lens, rp, cc, cv


def _run(ctx, M, x, y0, ref1, ref50, scale, what):
    """1 and 50 accumulating calls with a fresh host x every call (nothing of the caller's memory may stay mapped)"""
    y = y0.copy()
    ctx.apply_host(M, x.copy(), y)
    ol.assert_parity(y, ref1, scale, what + ", 1 call")
    for _ in range(NUM_TEST - 1):
        xs = x.copy()
        ctx.apply_host(M, xs, y)
        del xs
    ol.assert_parity(y, ref50, scale, what + ", 50 calls", reps=NUM_TEST)
    return y


@pytest.mark.parametrize("host_stores", ["1", "0"])
@pytest.mark.parametrize("shape", ["hub", "arrow"])
def test_apply_host_under_every_kernel_that_adds_into_y_with_atomics(pkg, orc, monkeypatch, shape, host_stores):
    capi = pkg.capi
    monkeypatch.setenv("SPMV_HOST_STORES", host_stores)  # read once per context
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    ctx = capi.Context(0)
    assert ctx.get_param("host_stores") in ((0, 1) if host_stores == "1" else (0,))
    rng = np.random.default_rng(41)
    n, lens, rp, cc, cv = (_hub if shape == "hub" else _arrow)(rng)
    assert 2 * n * 8 <= 1 << 20  # the staged path
    x = rng.uniform(0, 1, n)
    y0 = rng.uniform(-1, 1, n)
    prod, absprod = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, prod)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, absprod)
    ref1, ref50 = y0 + prod, y0 + NUM_TEST * prod
    scale = absprod + np.abs(y0) / NUM_TEST  # (assert_parity multiplies by reps)
    empty = np.flatnonzero(lens == 0)

    def check(M, what, atomics=None):
        if atomics is not None:
            assert M.get_param("adds_into_y_with_atomics") == atomics, what
        y = _run(ctx, M, x, y0, ref1, ref50, scale, f"{shape} host_stores={host_stores} {what}")
        assert np.array_equal(y[empty], y0[empty]), what  # rows without entries: untouched, bit for bit

    A = ctx.csr(n, n, rp, cc, cv)
    check(A, f"CSR AUTO (kernel {A.info.kernel})")
    forced = [(capi.CSR_VECTOR, None, 0), (capi.CSR_SCALAR, None, 0), (capi.CSR_PANEL, None, 0), (capi.CSR_TWOPHASE, None, 0),
              (capi.CSR_SEGSCAN, None, 1), (capi.CSR_SPLIT, 1, 1), (capi.CSR_SPLIT, 2, 0)]
    for kernel, mode, atomics in forced:
        if mode is not None:
            A.set_param("split_mode", mode)
        A.set_kernel(kernel)
        assert A.info.kernel == kernel
        if kernel == capi.CSR_SPLIT:
            assert A.get_param("split_long_rows") == 1 and A.get_param("split_mode") == mode
        check(A, f"CSR forced kernel {kernel} mode {mode}", atomics)
    A.set_param("split_mode", 0)
    # the long rows AND the short ones through atomics: every row with entries split off at threshold 1 leaves the inner copy
    # empty; at the default threshold the inner copy never picks the scan or a split of its own
    A.set_param("split_row_threshold", 1)
    A.set_param("split_mode", 1)
    A.set_kernel(capi.CSR_SPLIT)
    check(A, "CSR SPLIT, every row in chunks", 1)
    A.set_param("split_row_threshold", 0)
    A.set_param("split_mode", 0)
    try:
        A.set_kernel(capi.CSR_ELL)
    except capi.SpmvError as e:
        assert "out of proportion" in str(e) or "empty row" in str(e), e
    else:
        check(A, "CSR forced ELL copy", 0)
    del A
    rows = np.repeat(np.arange(n, dtype=np.int32), lens)
    # COO: the scan in place (atomics), the row-grouped copy (whatever kernel it picked), AUTO
    O = ctx.coo(n, n, rows, cc, cv)
    check(O, f"COO AUTO (kernel {O.info.kernel}, copy runs {O.get_param('rowgrouped_kernel')})")
    O.set_kernel(capi.CSR_VECTOR)
    check(O, "COO scan", 1)
    O.set_kernel(capi.CSR_PANEL)
    check(O, "COO row-grouped copy, panel", 0)
    del O
    # CSC: the scatter (atomics), the copy - AUTO lets the copy pick the scan or the split on these shapes (the round-5 hole)
    cp, cr, cw = ol.coo_to_csc(orc, n, rows, cc, cv)
    C = ctx.csc(n, n, cp, cr, cw)
    inner = C.get_param("rowgrouped_kernel")
    check(C, f"CSC AUTO (kernel {C.info.kernel}, copy runs {inner})")
    if C.info.kernel == capi.CSR_PANEL and inner == capi.CSR_SEGSCAN:
        assert C.get_param("adds_into_y_with_atomics") == 1
    C.set_kernel(capi.CSR_VECTOR)
    check(C, "CSC scatter", 1)
    C.set_kernel(capi.CSR_PANEL)
    check(C, "CSC row-grouped copy, panel", 0)
    del C
    ctx.close()


@pytest.mark.parametrize("host_stores", ["1", "0"])
def test_apply_host_sees_a_different_x_in_every_call(pkg, orc, monkeypatch, host_stores):
    """the CPU stores of x into device memory (large BAR) were only ever tested with the SAME x in all 50 calls - a
    stale cache line or an unflushed store would not have shown.  Here x changes in every call (and y accumulates), under a
    deterministic kernel: the result must equal, bit for bit, the resident product fed the same sequence through
    spmv_vec_upload - with the direct stores on and off."""
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.setenv("SPMV_HOST_STORES", host_stores)
    ctx = capi.Context(0)
    n, k = 10_000, 16  # C1's shape
    rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=3)
    A = ctx.csr(n, n, rp, cc, cv)
    A.set_kernel(capi.CSR_VECTOR, 4)  # a fixed tree per row
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-1, 1, n) for _ in range(12)]
    xs += [xs[0], np.zeros(n), xs[3]]  # ... and one seen before, zeros, another seen before
    dx, dy = ctx.vector(n), ctx.vector(n)
    dy.fill(0.0)
    y = np.zeros(n)
    for i, xv in enumerate(xs):
        dx.upload(xv)
        ctx.apply(A, dx, dy)
        ctx.sync()
        ctx.apply_host(A, xv.copy(), y)
        assert np.array_equal(y, dy.download()), f"call {i}: apply_host differs from the resident product (host_stores={host_stores})"
    ref, scale = np.zeros(n), np.zeros(n)
    for xv in xs:
        ol.csr_spmv(orc, rp, cc, cv, xv, ref)
        ol.csr_abs_row_sums(orc, rp, cc, cv, np.abs(xv), scale)
    ol.assert_parity(y, ref, scale, "apply_host over a changing x")
    if host_stores == "0":
        assert ctx.get_param("host_stores") == 0
    ctx.close()
