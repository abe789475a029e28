#!/usr/bin/env python3
"""tools/probe_twophase_again.py — ONE two-phase layout (values, columns, rows stay where they are), the placement search
of its product stream run again and again ("twophase_place_again", SPMV_TP_PLACEMENT_VERBOSE=1 prints every window and the
kept one timed again), the product timed the usual way after each: does a search's verdict hold for the product?"""
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_VERBOSE"] = "1"


def main():
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)

    def report(tag):
        out = []
        for only, reps in ((1, 10), (2, 10), (0, 2), (0, 10), (0, 50)):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, reps) for _ in range(3)))
        A.set_param("twophase_only", 0)
        print(f"{tag}: A {out[0]:.4f}  B {out[1]:.4f}  both: 2 products {out[2]:.4f}, 10 products {out[3]:.4f}, 50 products {out[4]:.4f} ms", flush=True)

    report("as built")
    held = []
    for i in range(6):
        if i % 2:
            held.append(ctx.vector((1 + i) * (1 << 27)))
        A.set_param("twophase_place_again", 1)
        report(f"search #{i}")
        time.sleep(0.5)
        report(f"search #{i}, 0.5 s later")


if __name__ == "__main__":
    main()
