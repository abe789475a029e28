"""GPU tests of the solver step around the product (spmv_apply_dot, spmv_cg).

The reference has no solver (its vec_dot / vec_axpby are never called), so there is no reference output to pin these
against ("parity unpinned" for the iteration itself).  What IS pinned: the y of apply_dot against the reference's golden
y (same gate as the product), its dot against the dot of that y, and the solution of CG through the oracle's product:
||b - A x|| / ||b|| computed on the CPU with oracle/spmv_oracle.c.
"""
import numpy as np
import pytest

import cases
import oracle_lib as ol
from conftest import golden
from test_gpu_parity import _csr_of, _scale

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_apply_dot_matches_golden_y_and_its_dot(ctx, orc, pkg, make):
    capi = pkg.capi
    c = make()
    g = golden(c["name"])
    rp, cc, cv = _csr_of(orc, c)
    scale = _scale(orc, c)
    w_host = np.random.default_rng(11).uniform(-1, 1, c["nrow"])
    mats = []
    for kernel in (capi.CSR_AUTO, capi.CSR_PANEL, capi.CSR_VECTOR, capi.CSR_SCALAR):
        A = ctx.csr(c["nrow"], c["ncol"], rp, cc, cv)
        A.set_kernel(kernel)
        mats.append((f"csr kernel={kernel}", A, "y1_csr"))
    mats.append(("coo", ctx.coo(c["nrow"], c["ncol"], ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"])), "y1_coo"))
    mats.append(("ell", ctx.coo_to_ell(mats[-1][1]), "y1_ell"))
    x, w = ctx.vector_from(c["x"]), ctx.vector_from(w_host)
    for name, A, key in mats:
        for overwrite in (True, False):
            y = ctx.vector(c["nrow"])
            y.fill(123.0 if overwrite else 0.0)  # overwrite must not see what was there
            d = ctx.apply_dot(A, x, y, w, overwrite=overwrite)
            got = y.download()
            ol.assert_parity(got, g[key], scale, f"{c['name']} apply_dot {name} overwrite={overwrite}")
            ref = float(np.dot(w_host, got))
            assert abs(d - ref) <= 1e-12 * float(np.dot(np.abs(w_host), np.abs(got))) + 1e-300, (name, overwrite, d, ref)


def _laplacian_2d(m):
    """5-point Laplacian on an m x m grid, CSR (symmetric positive definite)"""
    n = m * m
    idx = np.arange(n).reshape(m, m)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 4.0)]
    for a, b in ((idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((c, r))
    r, c, v = r[o], c[o], v[o]
    rp = np.zeros(n + 1, np.int32)
    np.add.at(rp, r + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), c.astype(np.int32), v


def _spd_random(n, k, seed):
    """B + B^T with a dominant diagonal: symmetric, strictly diagonally dominant -> positive definite"""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n), k)
    c = rng.integers(0, n, n * k)
    v = rng.uniform(-1, 1, n * k)
    keep = r != c
    r, c, v = r[keep], c[keep], v[keep]
    rr, cc, vv = np.concatenate([r, c]), np.concatenate([c, r]), np.concatenate([v, v])
    diag = np.zeros(n)
    np.add.at(diag, rr, np.abs(vv))
    rr, cc, vv = np.concatenate([rr, np.arange(n)]), np.concatenate([cc, np.arange(n)]), np.concatenate([vv, diag + 1.0])
    o = np.lexsort((cc, rr))
    rr, cc, vv = rr[o], cc[o], vv[o]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, rr + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), cc.astype(np.int32), vv


def _cg_numpy_iters(orc, rp, cc, cv, b, tol, max_iter):
    """the same recurrence on the CPU with the oracle's product, for the iteration count"""
    n = len(b)
    x, r = np.zeros(n), b.copy()
    p, rr = r.copy(), float(r @ r)
    bb = float(b @ b)
    for k in range(1, max_iter + 1):
        q = np.zeros(n)
        ol.csr_spmv(orc, rp, cc, cv, p, q)
        alpha = rr / float(p @ q)
        x += alpha * p
        r -= alpha * q
        rr_new = float(r @ r)
        if rr_new <= tol * tol * bb:
            return k
        p = r + (rr_new / rr) * p
        rr = rr_new
    return max_iter


@pytest.mark.parametrize("problem", ["laplacian", "random_spd", "random_spd_panel"])
def test_cg_solves_spd_systems(ctx, orc, pkg, problem):
    capi = pkg.capi
    if problem == "laplacian":
        n, rp, cc, cv = _laplacian_2d(96)
    elif problem == "random_spd":
        n, rp, cc, cv = _spd_random(20_000, 6, 5)
    else:
        n, rp, cc, cv = _spd_random(300_000, 5, 6)  # > 2M entries: the panel kernel with the fused dot
    A = ctx.csr(n, n, rp, cc, cv)
    if problem == "random_spd_panel":
        A.set_kernel(capi.CSR_PANEL)  # (AUTO times its candidates at this size: the test is about the panel kernel's fused dot)
        assert A.info.kernel == capi.CSR_PANEL
    b_host = np.random.default_rng(2).uniform(-1, 1, n)
    b, x = ctx.vector_from(b_host), ctx.vector(n)
    tol = 1e-9
    for check_every in (1, 7):
        x.fill(0.0)
        iters, relres = ctx.cg(A, b, x, max_iter=2000, rel_tol=tol, check_every=check_every)
        sol = x.download()
        ax = np.zeros(n)
        ol.csr_spmv(orc, rp, cc, cv, sol, ax)
        true_res = np.linalg.norm(b_host - ax) / np.linalg.norm(b_host)
        assert relres <= tol and true_res <= 20 * tol, (problem, check_every, iters, relres, true_res)
        if n <= 20_000:
            want = _cg_numpy_iters(orc, rp, cc, cv, b_host, tol, 2000)
            assert abs(iters - want) <= max(2, check_every), (problem, check_every, iters, want)
    # warm start from the solution: nothing left to do
    iters, relres = ctx.cg(A, b, x, max_iter=50, rel_tol=1e-6)
    assert iters == 0 and relres <= 1e-6


def test_cg_reports_a_matrix_that_is_not_positive_definite(ctx, pkg):
    n = 1000
    rp = np.arange(n + 1, dtype=np.int32)
    A = ctx.csr(n, n, rp, np.arange(n, dtype=np.int32), np.full(n, -1.0))  # -I
    b, x = ctx.vector_from(np.ones(n)), ctx.vector(n)
    x.fill(0.0)
    with pytest.raises(pkg.capi.SpmvError, match="not positive definite"):
        ctx.cg(A, b, x, max_iter=10, rel_tol=1e-8)
    rect = ctx.csr(n, n + 1, rp, np.arange(n, dtype=np.int32), np.ones(n))
    with pytest.raises(pkg.capi.SpmvError, match="not square"):
        ctx.cg(rect, b, x, max_iter=10, rel_tol=1e-8)


def test_cg_exact_convergence_between_host_checks_is_not_a_breakdown(ctx, pkg):
    """A = 2 I is solved exactly by the first step: r = 0, and the iterations queued up to the next look of the host
    (check_every = 16) run with p = 0, p.Ap = 0.  That is convergence, not `not positive definite`.  Same with a
    diagonal A under Jacobi (D^-1 A = I)."""
    n = 4096
    rp = np.arange(n + 1, dtype=np.int32)
    cc = np.arange(n, dtype=np.int32)
    b_host = np.random.default_rng(4).uniform(-1, 1, n)
    for diag, jacobi in ((np.full(n, 2.0), False), (np.random.default_rng(5).uniform(0.5, 50.0, n), True)):
        A = ctx.csr(n, n, rp, cc, diag)
        b, x = ctx.vector_from(b_host), ctx.vector(n)
        x.fill(0.0)
        iters, relres = ctx.cg(A, b, x, max_iter=64, rel_tol=1e-12, check_every=16, jacobi=jacobi)
        assert iters <= 16 and relres <= 1e-12, (jacobi, iters, relres)
        np.testing.assert_allclose(x.download(), b_host / diag, rtol=1e-14, atol=0)


def test_cg_never_reports_convergence_it_did_not_reach(ctx, pkg, monkeypatch):
    """Round 4's advisor: the two-launch iteration returned SPMV_OK with rel_resid = 0 and x = x0 (a) for -I under Jacobi
    (gamma = r.D^-1 r < 0: no status was set unless gamma > 0), (b) for a b with a NaN (the NaN r.r took the quiet
    noise-floor exit).  Both are errors in either arrangement of the iteration.  (c) Past the noise floor the attained
    residual is reported, not an exact 0."""
    n = 1000
    rp = np.arange(n + 1, dtype=np.int32)
    cc = np.arange(n, dtype=np.int32)
    for three in ("0", "1"):
        monkeypatch.setenv("SPMV_CG_THREE_LAUNCHES", three)
        A = ctx.csr(n, n, rp, cc, np.full(n, -1.0))  # -I: Jacobi's D^-1 is -I too
        b, x = ctx.vector_from(np.ones(n)), ctx.vector(n)
        for jacobi in (False, True):
            x.fill(0.0)
            with pytest.raises(pkg.capi.SpmvError, match="not positive definite"):
                ctx.cg(A, b, x, max_iter=10, rel_tol=1e-8, jacobi=jacobi)
        # NaN in b: an error, not iters = k with rel_resid = 0
        nn, rp2, cc2, cv2 = _laplacian_2d(40)
        L = ctx.csr(nn, nn, rp2, cc2, cv2)
        bad = np.ones(nn)
        bad[17] = np.nan
        for jacobi in (False, True):
            for check_every in (1, 8):
                bv, xv = ctx.vector_from(bad), ctx.vector(nn)
                xv.fill(0.0)
                with pytest.raises(pkg.capi.SpmvError, match="non-finite|not finite"):
                    ctx.cg(L, bv, xv, max_iter=40, rel_tol=1e-8, jacobi=jacobi, check_every=check_every)
        # NaN that enters later (x0 finite, b finite, a NaN matrix value in a row the first product multiplies by 0):
        cv3 = cv2.copy()
        cv3[rp2[5]] = np.nan
        Ln = ctx.csr(nn, nn, rp2, cc2, cv3)
        bv, xv = ctx.vector_from(np.ones(nn)), ctx.vector(nn)
        xv.fill(0.0)
        with pytest.raises(pkg.capi.SpmvError, match="non-finite|not finite|not positive definite"):
            ctx.cg(Ln, bv, xv, max_iter=40, rel_tol=1e-8, check_every=8)
    monkeypatch.setenv("SPMV_CG_THREE_LAUNCHES", "0")
    # (c) a tolerance below what fp64 can reach: iterations run out (or stop at the floor) and the residual reported is the
    # one attained (> 0 and small), not the cleared slot's 0
    nn, rp2, cc2, cv2 = _laplacian_2d(64)
    L = ctx.csr(nn, nn, rp2, cc2, cv2)
    b_host = np.random.default_rng(8).uniform(-1, 1, nn)
    bv, xv = ctx.vector_from(b_host), ctx.vector(nn)
    xv.fill(0.0)
    iters, relres = ctx.cg(L, bv, xv, max_iter=3000, rel_tol=1e-20, check_every=50)
    assert iters == 3000 and 0.0 < relres < 1e-12, (iters, relres)


def test_cg_graph_replay_takes_the_same_iterations(ctx, pkg, monkeypatch):
    """SPMV_CG_GRAPH=1 replays four captured iterations per launch.  Round 4's capture baked `k > 0 false` into its first node,
    so every replay restarted CG (beta = 0): valid iterates, slower convergence.  k now enters through k & 3 alone."""
    n, rp, cc, cv = _laplacian_2d(128)
    A = ctx.csr(n, n, rp, cc, cv)
    b_host = np.random.default_rng(3).uniform(-1, 1, n)
    counts = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("SPMV_CG_GRAPH", graph)
        for jacobi in (False, True):
            b, x = ctx.vector_from(b_host), ctx.vector(n)
            x.fill(0.0)
            iters, relres = ctx.cg(A, b, x, max_iter=4000, rel_tol=1e-9, check_every=8, jacobi=jacobi)
            assert relres <= 1e-9
            counts[(graph, jacobi)] = iters
    for jacobi in (False, True):
        assert abs(counts[("1", jacobi)] - counts[("0", jacobi)]) <= 8, counts


def test_sharded_cg_with_the_engine_as_local_ops():
    """dist.cg_sharded + dist.HipShardOps on one GPU (world 1; the N > 1 collectives are covered on CPU with gloo in
    tests/test_dist_gloo.py): same answer as the single-device spmv_cg.  In a child process, because torch must
    initialise its HIP runtime BEFORE the engine's library is loaded (as bench.py does) and this process has long
    loaded the engine."""
    import subprocess
    import sys
    from pathlib import Path

    child = Path(__file__).with_name("child_sharded_cg.py")
    r = subprocess.run([sys.executable, str(child)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED_CG_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_jacobi_preconditioned_cg(ctx, orc, pkg):
    """badly scaled SPD system S L S (L = 2-D Laplacian, S = diag of scales over six decades): plain CG crawls, the
    diagonal preconditioner undoes the scaling; both answers are checked through the oracle's product"""
    n, rp, cc, cv = _laplacian_2d(64)
    scale = 10.0 ** np.random.default_rng(8).uniform(-3, 3, n)
    rows = np.repeat(np.arange(n), np.diff(rp))
    cv = cv * scale[rows] * scale[cc]
    A = ctx.csr(n, n, rp, cc, cv)
    b_host = np.random.default_rng(9).uniform(-1, 1, n) * scale
    b, x = ctx.vector_from(b_host), ctx.vector(n)
    res = {}
    for jacobi in (False, True):
        x.fill(0.0)
        iters, relres = ctx.cg(A, b, x, max_iter=3000, rel_tol=1e-8, check_every=5, jacobi=jacobi)
        ax = np.zeros(n)
        ol.csr_spmv(orc, rp, cc, cv, x.download(), ax)
        res[jacobi] = (iters, relres, np.linalg.norm(b_host - ax) / np.linalg.norm(b_host))
    assert res[True][1] <= 1e-8 and res[True][2] <= 1e-6, res
    assert res[True][0] < 400 and res[True][0] * 3 < res[False][0], res  # ~the unscaled Laplacian's count vs thousands
    # error paths
    Err = pkg.capi.SpmvError
    rp0 = np.arange(4, dtype=np.int32)
    Z = ctx.csr(3, 3, rp0, np.array([1, 2, 0], np.int32), np.ones(3))  # no diagonal entries
    with pytest.raises(Err, match="diagonal"):
        ctx.cg(Z, ctx.vector_from(np.ones(3)), ctx.vector(3), jacobi=True)
    E = ctx.csr_to_ell(A)
    with pytest.raises(Err, match="CSR"):
        ctx.cg(E, b, x, jacobi=True)


# ---- symmetric Gauss-Seidel (spmv_symgs; SPMV_PRECOND_SYMGS) ------------------------------------------------------
def _laplacian_3d(m):
    """7-point Laplacian on an m^3 grid, CSR, columns ascending"""
    n = m * m * m
    idx = np.arange(n).reshape(m, m, m)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 6.0)]
    for a, b in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    o = np.lexsort((c, r))
    r, c, v = r[o], c[o], v[o]
    rp = np.zeros(n + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), c.astype(np.int32), v


def _dominant_random(n, k, seed, unsorted=True):
    """NON-symmetric pattern, k random off-diagonal entries per row in random order, the diagonal entry given twice
    (duplicates are summed), strictly diagonally dominant"""
    rng = np.random.default_rng(seed)
    cols = rng.integers(0, n, (n, k))
    vals = rng.uniform(-1, 1, (n, k))
    rows = np.repeat(np.arange(n), k).reshape(n, k)
    vals[cols == rows] = 0.0  # an accidental diagonal hit: keep the slot, drop its weight
    dom = np.abs(vals).sum(axis=1) + 1.0
    cc = np.concatenate([cols, rows[:, :1], rows[:, :1]], axis=1)
    cv = np.concatenate([vals, (0.75 * dom)[:, None], (0.25 * dom)[:, None]], axis=1)
    if unsorted:
        perm = np.argsort(rng.uniform(size=cc.shape), axis=1)
        cc, cv = np.take_along_axis(cc, perm, 1), np.take_along_axis(cv, perm, 1)
    rp = (np.arange(n + 1) * (k + 2)).astype(np.int32)
    return n, rp, cc.ravel().astype(np.int32), cv.ravel()


def _symgs_close(got, want, what):
    err = np.max(np.abs(got - want)) / max(np.max(np.abs(want)), 1e-300)
    assert err <= ol.REL_TOL, (what, err)


@pytest.mark.parametrize("order", [0, 1], ids=["row_order", "multicolour"])
@pytest.mark.parametrize("problem", ["tridiagonal_8", "laplacian_3d", "random_pattern", "lower_triangular"])
def test_symgs_matches_the_sweep_in_the_same_order(ctx, orc, pkg, problem, order):
    """spmv_symgs against oracle/spmv_oracle.c: orc_symgs_ordered (the textbook sweep; nothing in the reference to pin it
    to), in the matrix's own row order and in the multicolour order the engine reports: 1, 2 and 3 sweeps from a random
    x; the sequence against the oracle's sequential greedy colouring; the level structure the analysis found; the exact
    solution as a fixed point"""
    rng = np.random.default_rng(21)
    if problem == "tridiagonal_8":
        n = 8
        dense = 2.0 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
        r, c = np.nonzero(dense)
        rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
        cc, cv = c.astype(np.int32), dense[r, c]
        levels, colours = (n, n), 2  # row order: every row waits for its neighbour, a chain; even / odd rows
    elif problem == "laplacian_3d":
        m = 24
        n, rp, cc, cv = _laplacian_3d(m)
        levels, colours = (3 * m - 2, 3 * m - 2), 2  # hyperplanes i + j + k = const; red-black
    elif problem == "random_pattern":
        n, rp, cc, cv = _dominant_random(30_000, 7, 4)
        levels = colours = None
    else:
        n, rp, cc, cv = _dominant_random(5_000, 5, 6)
        rows = np.repeat(np.arange(n), np.diff(rp))
        keep = cc <= rows
        cc, cv = cc[keep], cv[keep]
        rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int32)
        levels = colours = None
    A = ctx.csr(n, n, rp, cc, cv)
    A.set_param("symgs_order", order)
    seq = ctx.symgs_order(A)
    assert A.get_param("symgs_order") == order
    if order == 0:
        assert np.array_equal(seq, np.arange(n)) and A.get_param("symgs_colours") == 0
    else:
        ncol, colour, want_seq = ol.greedy_colour_order(orc, rp, cc)
        assert A.get_param("symgs_colours") == ncol and np.array_equal(seq, want_seq)
        if colours:
            assert ncol == colours
    b_host, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b = ctx.vector_from(b_host)
    for sweeps in (1, 2, 3):
        want = x0.copy()
        assert ol.symgs(orc, rp, cc, cv, b_host, want, sweeps, order=seq) == 0
        x = ctx.vector_from(x0)
        ctx.symgs(A, b, x, sweeps)
        ctx.sync()
        _symgs_close(x.download(), want, f"{problem} order {order}: {sweeps} sweeps")
    lf, lb = A.get_param("symgs_levels_forward"), A.get_param("symgs_levels_backward")
    fused = A.get_param("symgs_fused")
    assert 1 <= lf <= n and 1 <= lb <= n and A.get_param("symgs_launches") >= (1 if fused else 4)
    if order == 1:
        # a proper colouring (no stored entry couples two rows of one colour) is swept with one launch per colour, forward
        # through all of them and back from the last but one; anything else takes the general scheme (levels inside colours)
        proper = not np.any((colour[np.repeat(np.arange(n), np.diff(rp))] == colour[cc]) & (np.repeat(np.arange(n), np.diff(rp)) != cc))
        assert bool(fused) == (proper and (lf, lb) == (ncol, ncol)), (problem, fused, proper, lf, lb, ncol)
        if fused:
            assert A.get_param("symgs_launches") == 2 * ncol - 1
        if problem in ("tridiagonal_8", "laplacian_3d"):
            assert fused == 1
        if problem == "random_pattern":
            assert fused == 0  # a_ij without a_ji: two coupled rows can share a colour
    else:
        assert fused == 0
    if levels and order == 0:
        assert (lf, lb) == levels, (lf, lb, levels)
    if colours and order == 1:
        assert (lf, lb) == (colours, colours), (lf, lb)  # a proper colouring: one level per colour
    if problem == "lower_triangular" and order == 0:
        # (L + D) x = b is solved by the forward half; the backward half has nothing above the diagonal to add
        assert lb == 1
        x = ctx.vector_from(x0)
        ctx.symgs(A, b, x, 1)
        ax = np.zeros(n)
        ol.csr_spmv(orc, rp, cc, cv, x.download(), ax)
        assert np.max(np.abs(ax - b_host)) <= 1e-12 * np.max(np.abs(b_host))
    # the solution of A x = b stays where it is: b := A x*
    xs = rng.uniform(-1, 1, n)
    bs = np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, xs, bs)
    x = ctx.vector_from(xs)
    ctx.symgs(A, ctx.vector_from(bs), x, 2)
    assert np.max(np.abs(x.download() - xs)) <= 1e-12
    # sweeps = 0 leaves x alone
    x = ctx.vector_from(x0)
    ctx.symgs(A, b, x, 0)
    assert np.array_equal(x.download(), x0)
    # the other order on the same handle: the plan is rebuilt
    A.set_param("symgs_order", 1 - order)
    seq2 = ctx.symgs_order(A)
    want = x0.copy()
    ol.symgs(orc, rp, cc, cv, b_host, want, 1, order=seq2)
    x = ctx.vector_from(x0)
    ctx.symgs(A, b, x, 1)
    _symgs_close(x.download(), want, f"{problem}: order switched to {1 - order}")


def test_symgs_preconditioned_cg_and_error_paths(ctx, orc, pkg):
    """CG on a 3-D Laplacian, plain / Jacobi / one symmetric Gauss-Seidel sweep per iteration: all three answers
    through the oracle's product; the sweep must save iterations (Jacobi cannot: the diagonal is constant)"""
    capi = pkg.capi
    n, rp, cc, cv = _laplacian_3d(40)
    A = ctx.csr(n, n, rp, cc, cv)
    b_host = np.random.default_rng(5).uniform(-1, 1, n)
    b, x = ctx.vector_from(b_host), ctx.vector(n)
    res = {}
    for name, kw in (("plain", {}), ("jacobi", {"jacobi": True}), ("symgs", {"symgs": True}), ("symgs_rows", {"symgs": True})):
        A.set_param("symgs_order", 0 if name == "symgs_rows" else 1)
        for check_every in (1, 6):
            x.fill(0.0)
            iters, relres = ctx.cg(A, b, x, max_iter=1000, rel_tol=1e-9, check_every=check_every, **kw)
            ax = np.zeros(n)
            ol.csr_spmv(orc, rp, cc, cv, x.download(), ax)
            true_res = np.linalg.norm(b_host - ax) / np.linalg.norm(b_host)
            assert relres <= 1e-9 and true_res <= 2e-8, (name, check_every, iters, relres, true_res)
            res[name, check_every] = iters
    assert res["symgs", 1] * 1.8 < res["plain", 1] and res["symgs_rows", 1] * 1.8 < res["plain", 1], res
    assert abs(res["jacobi", 1] - res["plain", 1]) <= 2, res
    assert res["symgs", 1] <= res["symgs", 6] <= res["symgs", 1] + 6, res
    # the same with the panel kernel doing the products
    A.set_param("symgs_order", 1)
    A.set_kernel(capi.CSR_PANEL)
    x.fill(0.0)
    iters, relres = ctx.cg(A, b, x, max_iter=1000, rel_tol=1e-9, symgs=True)
    assert relres <= 1e-9 and abs(iters - res["symgs", 1]) <= 1, (iters, res)
    # error paths
    Err = capi.SpmvError
    rp0 = np.arange(4, dtype=np.int32)
    Z = ctx.csr(3, 3, rp0, np.array([1, 2, 0], np.int32), np.ones(3))  # no diagonal entries
    with pytest.raises(Err, match="diagonal"):
        ctx.symgs(Z, ctx.vector_from(np.ones(3)), ctx.vector(3))
    with pytest.raises(Err, match="CSR"):
        ctx.symgs(ctx.csr_to_ell(A), b, x)
    R = ctx.csr(3, 4, rp0, np.array([0, 1, 2], np.int32), np.ones(3))
    with pytest.raises(Err, match="square"):
        ctx.symgs(R, ctx.vector_from(np.ones(3)), ctx.vector(3))
    with pytest.raises(Err, match="alias"):
        ctx.symgs(A, b, b)
