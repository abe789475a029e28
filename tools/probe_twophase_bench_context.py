#!/usr/bin/env python3
"""tools/probe_twophase_bench_context.py [REPS] — how often does the piece search (default budget) end on a fast stream when the
C5 shard is built the way bench.py builds it: after the headline matrix, the C3 / C4 / band extras have come and gone in the
same process (the allocator's free lists then hold their holes)?"""
import os
import statistics
import sys
import time
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
        for budget, offer in ((None, 0), (None, 1), (65536, 1)):
            t0 = time.perf_counter()
            M = ctx.gen_csr_uniform(7 * n, 8 * n, 8 * n, k, seed=1)
            if budget:
                M.set_param("twophase_placement_budget_mb", budget)
                M.set_param("twophase_choose_pieces", 1)
            M.set_param("twophase_offer_csr_copy", offer)  # round 5: the CSR copy's gigabytes offered to the search before they are released
            M.set_param("panel_keep_csr", 0)
            ctx.sync()
            setup = time.perf_counter() - t0
            vx, vy = ctx.gen_vector(8 * n, seed=1), ctx.vector(n)
            vy.fill(0.0)
            for _ in range(5):
                ctx.apply(M, vx, vy)
            ms = ctx.apply_timed(M, vx, vy, 50)
            out.append((r, (budget or 8192, offer), ms, M.get_param("twophase_pieces_exchanged"), M.get_param("twophase_placement_spread") / 1000))
            print(f"rep {r}, budget {budget or 8192} MB, CSR copy offered {offer}: {ms:.4f} ms, set-up {setup:.2f} s, configurations timed {M.get_param('twophase_placements_timed')}, "
                  f"pieces exchanged {out[-1][3]} (carved {M.get_param('twophase_pieces_carved')}), as built / kept {out[-1][4]:.3f}, bytes held {M.get_param('device_bytes') / 1e9:.2f} GB", flush=True)
            del M, vx, vy
    for b in ((8192, 0), (8192, 1), (65536, 1)):
        ts = [o[2] for o in out if o[1] == b]
        print(f"budget {b[0]} MB, CSR copy offered {b[1]}: " + " ".join(f"{t:.3f}" for t in ts) + f"; at or below 1.80 ms: {sum(t <= 1.80 for t in ts)} of {len(ts)}")


if __name__ == "__main__":
    main()
