#!/usr/bin/env python3
"""tools/tune.py — in-process A/B of kernel variants on the BASELINE configurations (GPU box only).

Interleaved rounds in ONE process (guide rule: perf deltas come from interleaved A/B, median + min).
    python tools/tune.py csr [--n 10000000 --k 32 --band 0]      lanes x flags sweep of the CSR kernels
    python tools/tune.py ell | coo                               C3 / C4
"""
import argparse
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402

capi = load_package().capi


def sweep(ctx, A, x, y, variants, rounds, reps, bytes_launch, nnz):
    res = {name: [] for name, _ in variants}
    for _ in range(rounds):
        for name, setup in variants:
            try:
                setup(A)
                ctx.apply(A, x, y)  # warm
                res[name].append(ctx.apply_timed(A, x, y, reps))
            except capi.SpmvError as e:
                print(f"# {name}: {e}")
                res[name].append(float("inf"))
    print(f"{'variant':28s} {'med ms':>9s} {'min ms':>9s} {'GFLOP/s':>9s} {'alg GB/s':>9s} {'%8TB/s':>7s}")
    for name, ts in res.items():
        med, mn = statistics.median(ts), min(ts)
        print(f"{name:28s} {med:9.4f} {mn:9.4f} {2*nnz/mn/1e6:9.1f} {bytes_launch/mn/1e6:9.1f} {bytes_launch/mn/1e6/80:7.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["csr", "ell", "coo", "blas1", "bandwin", "dia", "cg", "symgs"])
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--band", type=int, default=0)
    ap.add_argument("--ncol", type=int, default=0, help="csr: columns (default n); n x 8 = the shard shape of one rank of 8")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--full", action="store_true", help="every lanes x flags combination")
    ap.add_argument("--panel", default="", help="panel variants 'unroll,pipe,layout,sync[,rows];...' (0 / -1 = by trial)")
    ap.add_argument("--twophase", default="20000,6;20000,4;10000,6;10000,4", help="two-phase variants 'panel_cols,unroll;...'")
    ap.add_argument("--no-panel", action="store_true", help="csr: leave the panel variants out")
    a = ap.parse_args()
    ctx = capi.Context(0)
    if a.what == "blas1":
        import time

        n = a.n or 50_000_000
        x, y, w = ctx.gen_vector(n, seed=1), ctx.gen_vector(n, seed=2), ctx.vector(n)
        for name, (al, be), nbytes in (("axpby general", (0.5, 2.0), 24 * n), ("axpby alpha=1", (1.0, 2.0), 24 * n),
                                       ("axpby beta=0", (3.0, 0.0), 16 * n)):
            ctx.axpby(al, x, be, y, w)
            ctx.sync()
            t = time.perf_counter()
            for _ in range(20):
                ctx.axpby(al, x, be, y, w)
            ctx.sync()
            ms = (time.perf_counter() - t) / 20 * 1e3
            print(f"{name:16s} n={n}: {ms:.4f} ms  {nbytes / ms / 1e6:.1f} GB/s")
        ctx.dot(x, y)
        t = time.perf_counter()
        for _ in range(20):
            ctx.dot(x, y)  # synchronous (returns the scalar)
        ms = (time.perf_counter() - t) / 20 * 1e3
        print(f"{'dot':16s} n={n}: {ms:.4f} ms  {16 * n / ms / 1e6:.1f} GB/s (incl. result read-back)")
        return
    if a.what == "cg":
        # device-resident conjugate gradients (spmv_cg) on the 5-point Laplacian of an m x m grid
        import time

        import numpy as np

        m = a.n or 2048
        n = m * m
        idx = np.arange(n, dtype=np.int64).reshape(m, m)
        rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 4.0)]
        for u, v in ((idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])):
            rows += [u.ravel(), v.ravel()]
            cols += [v.ravel(), u.ravel()]
            vals += [np.full(u.size, -1.0)] * 2
        r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        o = np.lexsort((c, r))
        rp = np.zeros(n + 1, np.int64)
        np.add.at(rp, r + 1, 1)
        A = ctx.csr(n, n, np.cumsum(rp).astype(np.int32), c[o].astype(np.int32), v[o])
        nnz = len(v)
        b, x = ctx.gen_vector(n, seed=3), ctx.vector(n)
        print(f"CG, 5-point Laplacian {m} x {m}: n={n} nnz={nnz} kernel={A.info.kernel}")
        for iters_cap, every in ((200, 1), (200, 3), (200, 50), (200, 1), (200, 3), (200, 50)):  # twice: run-to-run spread
            x.fill(0.0)
            ctx.sync()
            t = time.perf_counter()
            it, res = ctx.cg(A, b, x, max_iter=iters_cap, rel_tol=0.0, check_every=every)
            dt = time.perf_counter() - t
            bytes_it = 12 * nnz + 4 * n + 16 * n + 48 * n + 24 * n  # product (A, p, q) + x/r update + direction
            print(f"  check_every={every:3d}: {it} iterations in {dt*1e3:.2f} ms = {dt/it*1e6:.1f} us/iteration, "
                  f"{bytes_it/(dt/it)/1e9:.0f} GB/s algorithmic, residual {res:.3e}")
        return
    if a.what == "symgs":
        # symmetric Gauss-Seidel (spmv_symgs) and CG preconditioned with it, 7-point Laplacian of an m^3 grid
        import time

        import numpy as np

        m = a.n or 160
        n = m * m * m
        idx = np.arange(n, dtype=np.int64).reshape(m, m, m)
        rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 6.0)]
        for u, v in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
            rows += [u.ravel(), v.ravel()]
            cols += [v.ravel(), u.ravel()]
            vals += [np.full(u.size, -1.0)] * 2
        r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
        o = np.lexsort((c, r))
        rp = np.zeros(n + 1, np.int64)
        np.add.at(rp, r + 1, 1)
        A = ctx.csr(n, n, np.cumsum(rp).astype(np.int32), c[o].astype(np.int32), v[o])
        nnz = len(v)
        b, x, y = ctx.gen_vector(n, seed=3), ctx.vector(n), ctx.vector(n)
        y.fill(0.0)
        prod = ctx.apply_timed(A, b, y, 20)
        print(f"SymGS, 7-point Laplacian {m}^3: n={n} nnz={nnz} kernel={A.info.kernel}, product {prod*1e3:.1f} us")
        for order in (1, 0):
            A.set_param("symgs_order", order)
            x.fill(0.0)
            ctx.sync()
            t = time.perf_counter()
            ctx.symgs(A, b, x, 0)  # the analysis alone
            ctx.sync()
            setup = time.perf_counter() - t
            print(f" order {order} ({'multicolour' if order else 'row order'}): set-up {setup*1e3:.1f} ms; {A.get_param('symgs_colours')} colours, levels "
                  f"{A.get_param('symgs_levels_forward')} / {A.get_param('symgs_levels_backward')}, {A.get_param('symgs_launches')} launches per sweep, "
                  f"{A.get_param('symgs_bytes')/1e6:.0f} MB")
            for sweeps in (1, 10, 10):
                ctx.sync()
                t = time.perf_counter()
                ctx.symgs(A, b, x, sweeps)
                ctx.sync()
                dt = (time.perf_counter() - t) / sweeps
                print(f"  {sweeps:2d} sweeps: {dt*1e3:.3f} ms per sweep = {dt*1e3/prod:.1f} products; "
                      f"{(2*12*nnz + 7*8*n) / dt / 1e9:.0f} GB/s algorithmic")
            x.fill(0.0)
            ctx.sync()
            t = time.perf_counter()
            it, res = ctx.cg(A, b, x, max_iter=2000, rel_tol=1e-8, check_every=10, symgs=True)
            dt = time.perf_counter() - t
            print(f"  CG + sweep: {it} iterations to {res:.2e} in {dt*1e3:.1f} ms = {dt/max(it,1)*1e6:.0f} us/iteration")
        for name, kw in (("plain", {}), ("jacobi", {"jacobi": True})):
            x.fill(0.0)
            ctx.sync()
            t = time.perf_counter()
            it, res = ctx.cg(A, b, x, max_iter=2000, rel_tol=1e-8, check_every=10, **kw)
            dt = time.perf_counter() - t
            print(f"  CG {name}: {it} iterations to {res:.2e} in {dt*1e3:.1f} ms = {dt/max(it,1)*1e6:.0f} us/iteration")
        return
    if a.what == "dia":
        n, k = a.n or 4_000_000, a.k or 64
        A = ctx.gen_dia_banded(n, k, seed=1)
        x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
        y.fill(0.0)
        # DIA streams 8 bytes per stored entry (no index array): values + x + y read/write
        sweep(ctx, A, x, y, [("dia tiled, x through LDS", lambda A: A.set_flags(0)), ("dia tiled, x from global memory", lambda A: A.set_flags(4))],
              a.rounds, a.reps, 8 * n * k + 8 * n + 16 * n, n * k)
        return
    if a.what == "bandwin":
        # non-wrapping band: rows [w, n-w) of the band matrix, so that every row block's window fits LDS
        import numpy as np

        n, k, band = a.n or 4_000_000, a.k or 32, a.band or 4096
        full = ctx.gen_csr_uniform(0, n, n, k, band=band, seed=1)
        rp, cc, cv = full.download()
        lo, hi = band, n - band
        A = ctx.csr(hi - lo, n, (rp[lo:hi + 1] - rp[lo]).astype(np.int32), cc[rp[lo]:rp[hi]], cv[rp[lo]:rp[hi]])
        del full
        x, y = ctx.gen_vector(n, seed=1), ctx.vector(hi - lo)
        y.fill(0.0)
        print(f"band {band}, rows {hi - lo}: auto kernel={A.info.kernel} window max={A.get_param('window_max_span')}")
        variants = [(f"ldswin L={l}", lambda A, l=l: A.set_kernel(capi.CSR_LDSWIN, l)) for l in (4, 8, 16)]
        variants += [("vector L=8", lambda A: A.set_kernel(capi.CSR_VECTOR, 8)), ("panel", lambda A: A.set_kernel(capi.CSR_PANEL))]
        nnz = A.info.nnz
        sweep(ctx, A, x, y, variants, a.rounds, a.reps, algorithmic_bytes("csr", hi - lo, n, nnz), nnz)
        return
    if a.what == "csr":
        n, k = a.n or 10_000_000, a.k or 32
        ncol = a.ncol or n
        A = ctx.gen_csr_uniform(0, n, ncol, k, band=a.band, seed=1)
        x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
        y.fill(0.0)
        info = A.info
        print(f"CSR n={n} ncol={ncol} k={k} band={a.band} auto: kernel={info.kernel} lanes={info.lanes_per_row}")
        variants = []
        for lanes in (2, 4, 8, 16, 32):
            for fl, tag in ((0, ""), (1, "+dpp"), (2, "+xcd"), (3, "+dpp+xcd")):
                def setup(A, lanes=lanes, fl=fl):
                    A.set_kernel(capi.CSR_VECTOR, lanes)
                    A.set_flags(fl)
                variants.append((f"vector L={lanes}{tag}", setup))
        if a.full:
            variants.append(("scalar", lambda A: A.set_kernel(capi.CSR_SCALAR)))
        else:
            variants = [v for v in variants if v[0] in ("vector L=8", "vector L=32", "vector L=8+xcd")]
        if a.panel:
            variants = []
            combos = [tuple(int(t) for t in item.split(",")) for item in a.panel.split(";")]
        else:
            combos = [(8, 2, 4, 1), (8, 2, 4, 3), (4, 2, 4, 3), (8, 1, 4, 1), (8, 1, 4, 3), (4, 2, 4, 1), (4, 1, 4, 0), (8, 1, 4, 0), (0, -1, 4, -1)]
        for combo in combos:  # --panel fields: unroll,pipe,layout,sync[,rows]  (0 / -1 = by trial)
            combo = combo + (0, -1, 4, -1, 0)[len(combo):]
            unroll, pipe, aos, sync, rows = combo[:5]

            def setup(A, unroll=unroll, pipe=pipe, aos=aos, sync=sync, rows=rows):
                for k, v in (("panel_aos", aos), ("panel_rows", rows), ("panel_unroll", unroll), ("panel_pipe", pipe), ("panel_sync", sync)):
                    A.set_param(k, v)
                A.set_kernel(capi.CSR_PANEL)  # rebuilds the layout when the parameters changed
            variants.append((f"panel U={unroll} pipe={pipe} layout={aos} sync={sync} rows={rows}", setup))
        if a.no_panel:
            variants = []
        for item in [t for t in a.twophase.split(";") if t]:
            pcols, unroll = (int(t) for t in item.split(","))

            def setup_tp(A, pcols=pcols, unroll=unroll):
                A.set_param("twophase_panel_cols", pcols)
                A.set_kernel(capi.CSR_TWOPHASE)  # rebuilds the layout when the panel width changed
            variants.append((f"two-phase cols={pcols} U={unroll}", setup_tp))
        if a.band and a.band <= 8192:
            for lanes in (4, 8, 16):
                variants.append((f"ldswin L={lanes}", lambda A, lanes=lanes: A.set_kernel(capi.CSR_LDSWIN, lanes)))
        sweep(ctx, A, x, y, variants, a.rounds, a.reps, algorithmic_bytes("csr", n, ncol, n * k), n * k)
        for name, v in (("panel_aos", 4), ("panel_rows", 0), ("panel_unroll", 0), ("panel_pipe", -1), ("panel_sync", -1)):
            A.set_param(name, v)
        A.set_kernel(capi.CSR_PANEL)
        chosen = {k: A.get_param("panel_" + k) for k in ("rows", "groups", "layout", "unroll", "pipe", "sync", "bytes")}
        print("chosen by trial:", chosen)
    elif a.what == "ell":
        n, k = a.n or 4_000_000, a.k or 64
        if a.band < 0:  # uniform-random columns: ELL made from the CSR generator
            A = ctx.csr_to_ell(ctx.gen_csr_uniform(0, n, n, k, seed=1))
        else:
            A = ctx.gen_ell_banded(n, n, k, seed=1)
        print(f"ELL n={n} k={k} {'uniform columns' if a.band < 0 else 'circulant band'}: auto kernel={A.info.kernel} (4 = panel copy, 1 = one lane per row)")
        x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
        y.fill(0.0)
        def x2(A, flags, lanes=2, tiled=0):
            A.set_flags(flags)
            A.set_kernel(capi.CSR_VECTOR, lanes)
            if A.get_param("ell_tiled_values") != tiled:
                A.set_param("ell_tiled_values", tiled)
        print("values in tiles of 512 rows:", A.get_param("ell_tiled_values"))
        variants = [("ell x2, slots as diagonals (no index stream)", lambda A: x2(A, 0)), ("the same, values in tiles of 512 rows", lambda A: x2(A, 0, 2, 1)),
                    ("the same, 8 slots in flight", lambda A: x2(A, 0, 4)),
                    ("the same, 2 slots in flight", lambda A: x2(A, 0, 8)), ("ell x2, column indices read", lambda A: x2(A, 8)),
                    ("the same, 8 slots in flight", lambda A: x2(A, 8, 4)), ("ell x1", lambda A: x2(A, 8, 1))]
        sweep(ctx, A, x, y, variants, a.rounds, a.reps, algorithmic_bytes("ell", n, n, n * k, k), n * k)
    else:
        n = a.n or 2_000_000
        A = ctx.gen_coo_powerlaw(n, n, 4096, seed=1)
        nnz = A.info.nnz
        print(f"COO power-law n={n} nnz={nnz} mean={nnz/n:.1f} sorted={A.info.sorted_rows}")
        x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
        y.fill(0.0)
        # AUTO regroups a large COO handle by row and runs the panel product; CSR_VECTOR forces the segmented scan
        sweep(ctx, A, x, y, [("coo auto (row-grouped, panel)", lambda A: A.set_kernel(capi.CSR_AUTO)),
                             ("coo segmented scan", lambda A: A.set_kernel(capi.CSR_VECTOR))],
              a.rounds, a.reps, algorithmic_bytes("coo", n, n, nnz), nnz)
        csr = ctx.coo_to_csr(A)
        print(f"same matrix as CSR: auto kernel={csr.info.kernel} lanes={csr.info.lanes_per_row} max_row={csr.info.max_row_nnz}")
        variants = [(f"csr vector L={l}", lambda A, l=l: A.set_kernel(capi.CSR_VECTOR, l)) for l in (16, 64)]
        for unroll, rows in ((0, 0), (8, 0), (4, 0), (0, n // 512 + 1), (0, n // 1024 + 1), (8, n // 512 + 1)):
            def setup(A, unroll=unroll, rows=rows):
                A.set_param("panel_unroll", unroll)
                A.set_param("panel_rows", rows)
                A.set_kernel(capi.CSR_PANEL)
            variants.append((f"csr panel U={unroll} rows={rows}", setup))
        sweep(ctx, csr, x, y, variants, a.rounds, a.reps, algorithmic_bytes("csr", n, n, nnz), nnz)
        print("panel layout:", {k: csr.get_param("panel_" + k) for k in ("rows", "groups", "unroll", "pipe", "sync", "layout")})


if __name__ == "__main__":
    main()
