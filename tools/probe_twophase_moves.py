#!/usr/bin/env python3
"""tools/probe_twophase_moves.py — one two-phase layout (no placement search); each of its streams is moved to fresh plain
allocations again and again ("twophase_realloc": 1 products, 2 values, 4 columns, 8 rows, 16 table), spacers held between
some moves; phase A and B alone after every move.  Which stream's placement decides the mode?"""
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
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)

    def phases():
        out = []
        for only in (1, 2):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, 10) for _ in range(3)))
        A.set_param("twophase_only", 0)
        return out

    a, b = phases()
    print(f"as built: A {a:.4f}  B {b:.4f}", flush=True)
    held = []
    for round_ in range(2):
        for name, bits in (("values", 2), ("columns", 4), ("table", 16), ("products", 1), ("rows", 8)):
            for rep, spacer in enumerate((0, 0, 1, 0, 2, 0, 3, 0)):
                if spacer:
                    held.append(ctx.vector(spacer * (1 << 27)))
                A.set_param("twophase_realloc", bits)
                a, b = phases()
                print(f"round {round_}, moved {name:9s} (#{rep}, spacer {spacer} GB): A {a:.4f}  B {b:.4f}", flush=True)
        held.clear()


if __name__ == "__main__":
    main()
