#!/usr/bin/env python3
"""tools/probe_twophase_placement.py — which stream's PLACEMENT decides the time of the two-phase product?

Round 2 found (profiles/r02_probe_twophase_placement.txt) that re-building the layout of the C5 shard shape in one
process gives the same device addresses and yet 1.82-1.88 or 1.95-2.03 ms per product.  Here single streams are moved to
fresh allocations (spmv_mat_set_param "twophase_realloc": bit 1 products, 2 values, 4 columns, 8 rows; contents copied,
the old allocation freed afterwards) and each phase is timed alone after every move."""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"  # "twophase_only" (one phase alone, wrong results) and "twophase_realloc" exist only with this


def main():
    n, ncol, k = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)

    def phases():
        out = []
        for only in (1, 2, 0):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, 5) for _ in range(3)))
        return out

    a, b, ab = phases()
    print(f"as built:                 A {a:.4f}  B {b:.4f}  both {ab:.4f}", flush=True)
    for name, bits in (("products", 1), ("values", 2), ("columns", 4), ("rows", 8)):
        for rep in range(5):
            A.set_param("twophase_realloc", bits)
            a, b, ab = phases()
            print(f"moved {name:9s} (#{rep}):   A {a:.4f}  B {b:.4f}  both {ab:.4f}", flush=True)


if __name__ == "__main__":
    main()
