"""Plans: a handle's set-up decisions exported, imported and shared (spmv_mat_get_plan / spmv_mat_set_plan / spmv_ctx_set_plan).

The reference builds its shards once and the same way every time (src/mat_vec.cpp:240-268); here AUTO is a measurement, so two
handles of one matrix may end on different kernels.  A plan pins the outcome: the handle built from it runs the same kernel in
the same layout with the same tuned parameters, with NO timing launch - checked through the handle's own record
("select_candidates" 0), through get_plan(set_plan(p)) == p, and through the products (within the parity gate for every kernel,
bit for bit for the kernels that add in a fixed order)."""
import struct

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
HEADER, NODE = 16, 128


def _nodes(plan):
    magic, version, nbytes, nnodes = struct.unpack_from("<IIII", plan, 0)
    assert magic == 0x4E4C5053 and version == 1 and nbytes == len(plan) == HEADER + NODE * nnodes
    return [struct.unpack_from("<32i", plan, HEADER + NODE * i) for i in range(nnodes)]


def _hub(rng, n=40_000):
    lens = np.full(n, 8, np.int64)
    lens[1234] = 30_000
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    cc = rng.integers(0, n, rp[-1]).astype(np.int32)
    cv = rng.uniform(-1, 1, rp[-1])
    return n, lens, rp, cc, cv


def _band(rng, n=300_000, half=3):
    i = np.repeat(np.arange(n, dtype=np.int64), 2 * half + 1)
    c = i + np.tile(np.arange(-half, half + 1, dtype=np.int64), n)
    ok = (c >= 0) & (c < n)
    rows, cols = i[ok], c[ok].astype(np.int32)
    lens = np.bincount(rows, minlength=n)
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    return n, lens, rp, cols, rng.uniform(-1, 1, rows.size)


def _product(ctx, M, dx, n):
    dy = ctx.vector(n)
    dy.fill(0.0)
    ctx.apply(M, dx, dy)
    ctx.sync()
    return dy.download()


def test_a_plan_rebuilds_every_csr_kernel_without_a_timing_launch(ctx, orc, pkg, monkeypatch):
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    rng = np.random.default_rng(61)
    n, lens, rp, cc, cv = _hub(rng)
    x = rng.uniform(0, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx = ctx.vector_from(x)
    A = ctx.csr(n, n, rp, cc, cv)
    assert A.get_param("select_candidates") >= 2  # AUTO timed its candidates
    auto_plan = A.get_plan()
    nodes = _nodes(auto_plan)
    assert nodes[0][0] == capi.FMT_CSR and nodes[0][1] == A.info.kernel
    B = ctx.csr(n, n, rp, cc, cv)
    B.set_plan(auto_plan)
    assert B.info.kernel == A.info.kernel and B.get_param("select_candidates") == 0 and B.get_plan() == auto_plan
    ol.assert_parity(_product(ctx, B, dx, n), ref, scale, f"hub row, the plan of AUTO (kernel {A.info.kernel})")
    setups = [(capi.CSR_VECTOR, 4, {}), (capi.CSR_VECTOR, 16, {}), (capi.CSR_SCALAR, 0, {}), (capi.CSR_SEGSCAN, 0, {}),
              (capi.CSR_PANEL, 0, {"panel_unroll": 4, "panel_pipe": 2, "panel_sync": 3, "panel_aos": 3}),
              (capi.CSR_PANEL, 0, {"panel_unroll": 8, "panel_pipe": 1, "panel_sync": 1, "panel_aos": 0, "panel_width": 65536}),
              (capi.CSR_TWOPHASE, 0, {"twophase_panel_cols": 8192}), (capi.CSR_SPLIT, 0, {"split_mode": 1}), (capi.CSR_SPLIT, 0, {"split_mode": 2}),
              (capi.CSR_SPLIT, 0, {"split_mode": 2, "split_row_threshold": 6})]
    for kernel, lanes, params in setups:
        S = ctx.csr(n, n, rp, cc, cv)
        for k, v in params.items():
            S.set_param(k, v)
        S.set_kernel(kernel, lanes)
        plan = S.get_plan()
        T = ctx.csr(n, n, rp, cc, cv)  # (selects by itself first: a plan replaces whatever a handle had)
        T.set_plan(plan)
        what = f"kernel {kernel} lanes {lanes} {params}"
        assert T.info.kernel == kernel and T.get_plan() == plan and T.get_param("select_candidates") == 0, what
        if lanes:
            assert T.info.lanes_per_row == lanes
        for k in ("panel_unroll", "panel_pipe", "panel_sync", "split_mode", "twophase_panel_cols"):
            if k in params:
                assert T.get_param(k) == S.get_param(k) == params[k], (what, k)
        if kernel == capi.CSR_PANEL:
            assert T.get_param("panel_layout") == S.get_param("panel_layout") and T.get_param("panel_groups") == S.get_param("panel_groups")
            assert T.get_param("panel_width") == S.get_param("panel_width") and T.get_param("panel_bytes") == S.get_param("panel_bytes")
        if kernel == capi.CSR_SPLIT:
            for k in ("split_row_threshold", "split_long_rows", "split_virtual_rows", "split_inner_kernel", "split_long_kernel"):
                assert T.get_param(k) == S.get_param(k), (what, k)
            assert len(_nodes(plan)) == (3 if params["split_mode"] == 2 else 2)
        ys, yt = _product(ctx, S, dx, n), _product(ctx, T, dx, n)
        ol.assert_parity(yt, ref, scale, "from a plan: " + what)
        if kernel in (capi.CSR_VECTOR, capi.CSR_SCALAR):  # a fixed order per row: two handles of one plan give the same bits
            assert np.array_equal(ys, yt), what
        assert T.get_param("device_bytes") <= S.get_param("device_bytes"), what  # the plan's layout and nothing left over from the handle's own AUTO


def test_plans_carry_the_copies_a_handle_runs_from(ctx, orc, pkg, monkeypatch):
    """COO / CSC / ELL handles that run from a row-grouped copy, a CSR handle that runs from its ELL copy: the plan holds a node
    per copy, and a handle built from it runs the same copy with the same inner kernel"""
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    rng = np.random.default_rng(67)
    n, lens, rp, cc, cv = _hub(rng)
    x = rng.uniform(0, 1, n)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, x, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, x, scale)
    dx = ctx.vector_from(x)
    rows = np.repeat(np.arange(n, dtype=np.int32), lens)
    cp, cr, cw = ol.coo_to_csc(orc, n, rows, cc, cv)
    makers = {"coo": lambda: ctx.coo(n, n, rows, cc, cv), "csc": lambda: ctx.csc(n, n, cp, cr, cw)}
    for fmt, make in makers.items():
        for forced in (None, capi.CSR_VECTOR, capi.CSR_PANEL):
            S = make()
            if forced is not None:
                S.set_kernel(forced)
            plan = S.get_plan()
            # (one node for the handle, one for its copy - and more where the copy runs from copies of its own: a split's parts)
            assert (len(_nodes(plan)) >= 2) == (S.info.kernel == capi.CSR_PANEL)
            T = make()
            T.set_plan(plan)
            what = f"{fmt} forced {forced}: kernel {S.info.kernel}, copy runs {S.get_param('rowgrouped_kernel')}"
            assert T.info.kernel == S.info.kernel and T.get_param("rowgrouped_kernel") == S.get_param("rowgrouped_kernel"), what
            assert T.get_plan() == plan and T.get_param("select_candidates") == 0, what
            assert T.get_param("adds_into_y_with_atomics") == S.get_param("adds_into_y_with_atomics")
            ol.assert_parity(_product(ctx, T, dx, n), ref, scale, "from a plan: " + what)
    # an ELL handle of few long rows: AUTO's trial ends on the row-grouped copy; a plan of the one-row-per-lane variant too
    nr, K, ncol = 3000, 96, 400_000
    ec = rng.integers(0, ncol, nr * K).astype(np.int32)
    ev = rng.uniform(-1, 1, nr * K)
    xe = rng.uniform(0, 1, ncol)
    refe = np.zeros(nr)
    ol.ell_spmv(orc, nr, K, ec, ev, xe, refe, fma=True)
    dxe = ctx.vector_from(xe)
    for forced, lanes in ((None, 0), (capi.CSR_PANEL, 0), (capi.CSR_VECTOR, 1), (capi.CSR_VECTOR, 2)):
        S = ctx.ell(nr, ncol, K, nr * K, ec, ev)
        if forced is not None:
            S.set_kernel(forced, lanes)
        plan = S.get_plan()
        T = ctx.ell(nr, ncol, K, nr * K, ec, ev)
        T.set_plan(plan)
        assert T.info.kernel == S.info.kernel and T.get_plan() == plan
        if S.info.kernel == capi.CSR_VECTOR:  # (a handle that runs from its copy has no variant in effect: plans are canonical)
            assert T.get_param("ell_variant") == S.get_param("ell_variant")
        assert T.get_param("rowgrouped_kernel") == S.get_param("rowgrouped_kernel") and T.get_param("select_candidates") == 0
        got = _product(ctx, T, dxe, nr)
        assert np.max(np.abs(got - refe)) <= ol.REL_TOL * K
        if T.info.kernel == capi.CSR_VECTOR:
            assert np.array_equal(got, refe)  # the format's own kernels add in the reference's order
    # a band: the CSR handle's ELL copy, diagonal slots and all
    nb, lb, rpb, cb, vb = _band(rng)
    xb = rng.uniform(0, 1, nb)
    refb, scb = np.zeros(nb), np.zeros(nb)
    ol.csr_spmv(orc, rpb, cb, vb, xb, refb)
    ol.csr_abs_row_sums(orc, rpb, cb, vb, xb, scb)
    S = ctx.csr(nb, nb, rpb, cb, vb)
    S.set_kernel(capi.CSR_ELL)
    plan = S.get_plan()
    assert len(_nodes(plan)) == 2 and _nodes(plan)[1][0] == capi.FMT_ELL
    T = ctx.csr(nb, nb, rpb, cb, vb)
    T.set_plan(plan)
    assert T.info.kernel == capi.CSR_ELL and T.get_plan() == plan and T.get_param("ell_copy_variant") == S.get_param("ell_copy_variant")
    assert T.get_param("ell_copy_diagonal_slots") == S.get_param("ell_copy_diagonal_slots") == 1
    dxb = ctx.vector_from(xb)
    yt = _product(ctx, T, dxb, nb)
    ol.assert_parity(yt, refb, scb, "band, ELL copy from a plan")
    assert np.array_equal(yt, _product(ctx, S, dxb, nb))  # the ELL kernels add in a fixed order: the same bits


def test_a_context_plan_is_taken_by_the_handles_created_under_it(pkg, orc, monkeypatch):
    """spmv_ctx_set_plan: uploads, generators, conversions and extracted shards of the plan's format take it instead of selecting;
    other formats and handles created after it was cleared select as usual.  This is how ranks of one job share rank 0's
    decisions (dist.broadcast_plan) and how a re-uploaded container keeps its set-up (compat.cpp)."""
    capi, synth = pkg.capi, pkg.synth
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    ctx = capi.Context(0)
    n, k = 200_000, 16  # 3.2M entries: AUTO times its candidates here
    A = ctx.gen_csr_uniform(0, n, n, k, seed=5)
    assert A.get_param("select_candidates") >= 2
    A.set_param("panel_unroll", 4)
    A.set_param("panel_sync", 3)
    A.set_kernel(capi.CSR_PANEL)  # (a configuration AUTO would not end on by itself: what follows can only come from the plan)
    plan = A.get_plan()
    ctx.set_plan(plan)
    rp, cc, cv = A.download()
    made = {"generator": ctx.gen_csr_uniform(0, n, n, k, seed=5), "upload": ctx.csr(n, n, rp, cc, cv),
            "shard of another size": ctx.extract_rows(A, 1000, 150_000), "another matrix": ctx.gen_csr_uniform(0, 90_000, n, 24, seed=9)}
    rows = np.repeat(np.arange(n, dtype=np.int32), k)
    C = ctx.coo(n, n, rows, cc, cv)
    made["conversion"] = ctx.coo_to_csr(C)
    for what, M in made.items():
        assert M.info.kernel == capi.CSR_PANEL and M.get_param("select_candidates") == 0, what
        assert M.get_param("panel_unroll") == 4 and M.get_param("panel_sync") == 3 and M.get_plan() == plan, what
    # the COO handle is not the plan's format: it selected as usual (and its copy did not take the context's plan either)
    assert C.get_param("select_candidates") == 2
    # products: the generator's twin against the handle the plan came from, the shard against its rows
    x = ctx.gen_vector(n, seed=5)
    hx = x.download()
    y_a, y_g = _product(ctx, A, x, n), _product(ctx, made["generator"], x, n)
    assert np.max(np.abs(y_a - y_g)) <= ol.REL_TOL * k
    ys = _product(ctx, made["shard of another size"], x, 149_000)
    assert np.max(np.abs(ys - y_a[1000:150_000])) <= ol.REL_TOL * k
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, hx, ref)
    ol.csr_abs_row_sums(orc, rp, cc, cv, hx, scale)
    ol.assert_parity(y_g, ref, scale, "generator under a context plan")
    ctx.set_plan(None)
    D = ctx.gen_csr_uniform(0, n, n, k, seed=5)
    assert D.get_param("select_candidates") >= 2  # back to selecting
    del made, A, C, D
    ctx.close()


def test_blobs_are_checked_before_anything_is_read_from_them(ctx, pkg):
    capi = pkg.capi
    rng = np.random.default_rng(71)
    n, lens, rp, cc, cv = _hub(rng, 20_000)
    A = ctx.csr(n, n, rp, cc, cv)
    A.set_param("split_mode", 2)
    A.set_kernel(capi.CSR_SPLIT)
    plan = A.get_plan()
    assert len(_nodes(plan)) == 3
    bad = {
        "magic": b"XXXX" + plan[4:], "version": plan[:4] + struct.pack("<I", 99) + plan[8:], "truncated": plan[:-1], "empty": b"",
        "node count": plan[:12] + struct.pack("<I", 200) + plan[16:],
        "child index backwards": plan[:HEADER + 4 * 20] + struct.pack("<i", 0) + plan[HEADER + 4 * 21:],
        "child index out of range": plan[:HEADER + 4 * 20] + struct.pack("<i", 7) + plan[HEADER + 4 * 21:],
        "kernel id": plan[:HEADER + 4] + struct.pack("<i", 42) + plan[HEADER + 8:],
        "format": plan[:HEADER] + struct.pack("<i", 9) + plan[HEADER + 4:],
        "lanes": plan[:HEADER + 8] + struct.pack("<i", 3) + plan[HEADER + 12:],
    }
    for what, blob in bad.items():
        with pytest.raises(capi.SpmvError, match="plan"):
            A.set_plan(blob)
        assert A.info.kernel == capi.CSR_SPLIT, what  # refused before the handle was touched
        if blob:
            with pytest.raises(capi.SpmvError, match="plan"):
                ctx.set_plan(blob)
    # a plan for another format
    rows = np.repeat(np.arange(n, dtype=np.int32), lens)
    C = ctx.coo(n, n, rows, cc, cv)
    with pytest.raises(capi.SpmvError, match="format"):
        C.set_plan(plan)
    # a plan that does not fit the matrix: the LDS-window kernel on a matrix whose row blocks span every column.  The call
    # fails, the handle selects by itself and still multiplies
    bn, bl, brp, bc, bv = _band(rng, 100_000, 2)
    B = ctx.csr(bn, bn, brp, bc, bv)
    B.set_kernel(capi.CSR_LDSWIN)
    lds_plan = B.get_plan()
    with pytest.raises(capi.SpmvError, match="does not fit"):
        A.set_plan(lds_plan)
    assert A.info.kernel != capi.CSR_LDSWIN
    x = rng.uniform(0, 1, n)
    got = _product(ctx, A, ctx.vector_from(x), n)
    ref = np.bincount(rows, weights=cv * x[cc], minlength=n)
    assert np.max(np.abs(got - ref)) <= 1e-10 * np.max(np.abs(ref))


def test_a_two_phase_handle_built_from_a_plan_can_still_run_its_piece_search(pkg, monkeypatch):
    """What `bench.py --gpus N` does on every rank but the first: the shard is built under rank 0's plan (two-phase layout, no
    timing launch, no piece search), and the search over where the product stream lies in THIS device's memory - no part of a
    plan - is run afterwards with the budget the job grants ("twophase_placement_budget_mb" + "twophase_choose_pieces")."""
    capi = pkg.capi
    monkeypatch.delenv("SPMV_PANEL_TRIAL", raising=False)
    monkeypatch.delenv("SPMV_TP_PLACEMENT_BUDGET_MB", raising=False)
    ctx = capi.Context(0)
    n, ncol, k = 4_500_000, 72_000_000, 16  # 72M entries: a product stream of 0.6 GB (searched from 512 MB on), x 16x the rows
    A = ctx.gen_csr_uniform(0, n, ncol, k, seed=3)
    assert A.info.kernel == capi.CSR_TWOPHASE
    plan = A.get_plan()
    x = ctx.gen_vector(ncol, seed=3)
    ya, yb = ctx.vector(n), ctx.vector(n)
    ya.fill(0.0)
    ctx.apply(A, x, ya)
    ctx.sync()
    ref = ya.download()
    del A, ya
    ctx.set_plan(plan)
    B = ctx.gen_csr_uniform(0, n, ncol, k, seed=3)
    ctx.set_plan(None)
    assert B.info.kernel == capi.CSR_TWOPHASE and B.get_plan() == plan
    assert B.get_param("select_candidates") == 0 and B.get_param("twophase_placements_timed") == 0  # nothing was timed, nothing searched
    B.set_param("twophase_placement_budget_mb", 3072)
    B.set_param("twophase_choose_pieces", 1)
    assert B.get_param("twophase_placements_timed") > 0 and B.get_param("twophase_placement_spread") >= 1000
    assert B.get_plan() == plan  # where the pieces lie is no part of the plan
    yb.fill(0.0)
    ctx.apply(B, x, yb)
    ctx.sync()
    assert np.max(np.abs(yb.download() - ref)) <= ol.REL_TOL * k * np.max(np.abs(ref))
    B.set_param("panel_keep_csr", 0)  # ... and what the bench does next: the CSR copy goes (its gigabytes offered to the search first)
    yb.fill(0.0)
    ctx.apply(B, x, yb)
    ctx.sync()
    assert np.max(np.abs(yb.download() - ref)) <= ol.REL_TOL * k * np.max(np.abs(ref))
    del B
    ctx.close()
