#!/usr/bin/env python3
"""tools/probe_twophase_bench_context.py [REPS] — how often does the piece search (default budget) end on a fast stream when the
C5 shard is built the way bench.py builds it: after the headline matrix, the C3 / C4 / band extras have come and gone in the
same process (the allocator's free lists then hold their holes)?"""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    n, k = 10_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, n, k, seed=1)  # the headline matrix stays resident, as in bench.py
    A.set_param("panel_keep_csr", 0)
    x1, y1 = ctx.gen_vector(n, seed=1), ctx.vector(n)
    out = []
    for r in range(reps):
        for make, ncol in ((lambda: ctx.gen_ell_banded(4_000_000, 4_000_000, 64, seed=1), 4_000_000),
                           (lambda: ctx.gen_coo_powerlaw(2_000_000, 2_000_000, 4096, seed=1), 2_000_000),
                           (lambda: ctx.gen_csr_uniform(0, n, n, k, band=65536, seed=1), n)):
            M = make()
            vx, vy = ctx.gen_vector(ncol, seed=1), ctx.vector(M.info.nrow)
            vy.fill(0.0)
            ctx.apply(M, vx, vy)
            ctx.sync()
            del M, vx, vy
        for budget in (None, 24576):
            M = ctx.gen_csr_uniform(7 * n, 8 * n, 8 * n, k, seed=1)
            if budget:
                M.set_param("twophase_placement_budget_mb", budget)
                M.set_param("twophase_choose_pieces", 1)
            M.set_param("panel_keep_csr", 0)
            vx, vy = ctx.gen_vector(8 * n, seed=1), ctx.vector(n)
            vy.fill(0.0)
            for _ in range(5):
                ctx.apply(M, vx, vy)
            ms = ctx.apply_timed(M, vx, vy, 50)
            out.append((r, budget or 8192, ms, M.get_param("twophase_pieces_exchanged"), M.get_param("twophase_placement_spread") / 1000))
            print(f"rep {r}, budget {budget or 8192} MB: {ms:.4f} ms, pieces exchanged {out[-1][3]}, as built / kept {out[-1][4]:.3f}", flush=True)
            del M, vx, vy
    for b in (8192, 24576):
        ts = [o[2] for o in out if o[1] == b]
        print(f"budget {b}: " + " ".join(f"{t:.3f}" for t in ts) + f"; at or below 1.80 ms: {sum(t <= 1.80 for t in ts)} of {len(ts)}")


if __name__ == "__main__":
    main()
