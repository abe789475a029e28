#!/usr/bin/env python3
"""tools/probe_twophase_held.py MODE COUNT — COUNT product streams of one two-phase layout held AT THE SAME TIME (allocator MODE:
0 hipMalloc, 1 one mapped piece, 2 mapped 1 GB pieces), phase A / B timed on each in turn, three passes: is the mode a
property of the memory a stream occupies (stable per stream, differing between streams held together)?"""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_TRIES"] = "1"


def main():
    mode, count = int(sys.argv[1]), int(sys.argv[2])
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

    for _ in range(count - 1):
        A.set_param("twophase_products_push", mode)
    for p in range(3):
        line = []
        for i in range(count):
            a, b = phases()
            line.append(f"{a:.3f}/{b:.3f}")
            A.set_param("twophase_products_rotate", 1)
        print(f"pass {p}: A/B per held stream (the stream built with the layout comes last): " + "  ".join(line), flush=True)


if __name__ == "__main__":
    main()
