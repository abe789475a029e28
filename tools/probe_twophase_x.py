#!/usr/bin/env python3
"""tools/probe_twophase_x.py — is it the placement (or the content) of x, not of the product stream, that decides phase A's
mode?  One two-phase layout of the C5 shard, built once without a search; phase A timed with several x vectors: the same
content at different addresses (spacers between the allocations), zeros, and the same vector again at the end."""
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
    assert A.info.kernel == capi.CSR_TWOPHASE
    y = ctx.vector(n)
    y.fill(0.0)

    def phases(x):
        out = []
        for only in (1, 2):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, 10) for _ in range(3)))
        A.set_param("twophase_only", 0)
        return out

    xs, held = [], []
    for i, spacer_gb in enumerate((0, 0, 1, 2, 3, 5, 0, 7)):
        if spacer_gb:
            held.append(ctx.vector(spacer_gb * (1 << 27)))
        x = ctx.gen_vector(ncol, seed=1)
        xs.append(x)
        a, b = phases(x)
        print(f"x #{i} at {x.device_ptr:#x} (spacer {spacer_gb} GB before it): A {a:.4f}  B {b:.4f}", flush=True)
    for i in (0, 3, 5):
        a, b = phases(xs[i])
        print(f"x #{i} again: A {a:.4f}  B {b:.4f}", flush=True)
    for i in (0, 3, 5):
        xs[i].fill(0.0)
        a, b = phases(xs[i])
        print(f"x #{i} filled with zeros: A {a:.4f}  B {b:.4f}", flush=True)
    for i in (0, 3, 5):
        xs[i].fill(0.75)
        a, b = phases(xs[i])
        print(f"x #{i} filled with 0.75: A {a:.4f}  B {b:.4f}", flush=True)
    # and y somewhere else
    for i in range(3):
        held.append(ctx.vector((i + 1) * (1 << 27)))
        y = ctx.vector(n)
        y.fill(0.0)
        a, b = phases(xs[1])
        print(f"x #1, y moved (#{i}): A {a:.4f}  B {b:.4f}", flush=True)
    for rep in range(4):
        A.set_param("twophase_realloc", 1)
        a, b = phases(xs[1])
        print(f"x #1, product stream moved (#{rep}): A {a:.4f}  B {b:.4f}", flush=True)
        a, b = phases(xs[2])
        print(f"x #2, same stream: A {a:.4f}  B {b:.4f}", flush=True)


if __name__ == "__main__":
    main()
