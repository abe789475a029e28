#!/usr/bin/env python3
"""tools/probe_twophase_counters.py — run under `rocprofv3 --pmc <TCC counters> --output-format csv`: the product stream of the
C5 shard's two-phase layout is moved to fresh allocations (spacers between the moves) and phase A is launched three times after
every move, so that the per-dispatch counters of fast and slow placements can be compared in ONE process."""
import os
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
    held = []
    for move, spacer_gb in enumerate((0, 0, 1, 2, 4, 4, 8, 8, 2, 1, 0, 4)):
        if spacer_gb:
            held.append(ctx.vector(spacer_gb * (1 << 27)))
        if move:
            A.set_param("twophase_realloc", 1)
        for _ in range(3):
            ctx.apply(A, x, y)
        ctx.sync()
        print(f"move {move}: spacer {spacer_gb} GB", flush=True)


if __name__ == "__main__":
    main()
