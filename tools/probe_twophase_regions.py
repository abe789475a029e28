#!/usr/bin/env python3
"""tools/probe_twophase_regions.py — how far apart in the allocator's order do placements of the two-phase product stream
have to be to differ in mode?  The stream is moved to a fresh allocation (twophase_realloc bit 1) again and again; between
moves a spacer of S GB is allocated and HELD, so that the next candidate comes from memory further along.  Prints phase A's
time after every move."""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_BUDGET_MB"] = "0"  # no piece search: the pieces the allocator hands out


def main():
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    A.set_param("twophase_only", 1)

    def phase_a():
        ctx.apply(A, x, y)
        return statistics.median(ctx.apply_timed(A, x, y, 5) for _ in range(3))

    print(f"as built: A {phase_a():.4f} ms", flush=True)
    held = []
    for spacer_gb in (0, 0, 0, 1, 1, 2, 2, 4, 4, 8, 8, 16, 16, 0, 0, 32, 0, 0):
        if spacer_gb:
            held.append(ctx.vector(spacer_gb * (1 << 27)))  # doubles: 2^27 * 8 B = 1 GB
        A.set_param("twophase_realloc", 1)
        free, total = ctx.mem_info()
        print(f"spacer {spacer_gb:2d} GB held before this candidate (free {free / 2**30:.0f} GB): A {phase_a():.4f} ms", flush=True)


if __name__ == "__main__":
    main()
