#!/usr/bin/env python3
"""tools/sweep_panel_configs.py - every (chunk size, order, barrier placement) of the panel product on a handful of matrices, same
box, same layout in memory.  ms = min of 3 x apply_timed(20).  What the build-time trial (panel_choose_pace) should have on its
list.  (Round 5 ran it with a second column - the round 2-4 kernel through the A/B switch "panel_legacy" - before that kernel and
the switch were deleted: profiles/r05_sweep_panel_configs_written_down_vs_round4.txt.)"""
import itertools
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
ctx = capi.Context(0)
ORDER = {0: "no prefetch", 1: "stream-first", 2: "gather-first"}


def sweep(name, A, ncol, aos=None):
    if aos is not None:
        A.set_param("panel_aos", aos)
        A.set_kernel(4)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(int(A.info.nrow))
    y.fill(0.0)
    chosen = {k: A.get_param("panel_" + k) for k in ("layout", "unroll", "pipe", "sync")}
    print(f"== {name}: the trial chose {chosen}", flush=True)
    rows = []
    for u, order, sync in itertools.product((4, 8), (1, 2), (0, 1, 3)):
        A.set_param("panel_unroll", u)
        A.set_param("panel_pipe", order)
        A.set_param("panel_sync", sync)
        A.set_kernel(4)
        ctx.apply(A, x, y)
        rows.append((min(ctx.apply_timed(A, x, y, 20) for _ in range(3)), u, order, sync))
    best = min(r[0] for r in rows)
    for t, u, order, sync in sorted(rows):
        print(f"   U = {u}  {ORDER[order]:12s} sync {sync}: {t:.4f} ms{'   <- best' if t == best else ''}", flush=True)
    for k in ("unroll", "pipe", "sync"):
        A.set_param("panel_" + k, 0 if k == "unroll" else -1)
    A.set_kernel(4)


n = 10_000_000
A = ctx.gen_csr_uniform(0, n, n, 32, seed=1)
sweep("C2: 10M x 10M x 32 uniform, packed paired slices", A, n)
sweep("C2, three-array layout", A, n, aos=0)
del A
A = ctx.gen_csr_uniform(0, n, 2 * n, 32, seed=1)
sweep("N = 2 shard shape 10M x 20M", A, 2 * n)
del A
A = ctx.gen_csr_uniform(0, n, n, 32, band=65536, seed=1)
sweep("band 65536", A, n)
del A
A = ctx.gen_csr_uniform(0, n, n, 32, band=4096, seed=1)
sweep("band 4096", A, n)
del A
C = ctx.gen_coo_powerlaw(2_000_000, 2_000_000, 4096, seed=1)
R = ctx.coo_to_csr(C)
del C
R.set_kernel(4)
sweep("C4 grouped by row", R, 2_000_000)
