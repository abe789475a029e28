#!/usr/bin/env python3
"""tools/probe_twophase_placement.py — the two-phase product of the C5 shard shape, its layout built several times in
one process (alternating two panel widths forces a re-build: every stream is freed and allocated again).  Measured
(profiles/r02_probe_twophase_placement.txt): the streams come back at the SAME device addresses, filling the same
memory again with the entries in another order changes nothing, and yet a re-build lands anywhere from 1.79 to 2.03 ms
per product (phase A: 1.22 or 1.38 ms) — what differs is the physical memory behind the addresses."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    n, ncol, k = 10_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 80_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    for build in range(8):
        for cols in (10_000, 20_000):  # the first forces the re-build of the second
            A.set_param("twophase_panel_cols", cols)
            A.set_kernel(capi.CSR_TWOPHASE)
        ts = []
        for _ in range(3):
            ctx.apply(A, x, y)
            ts.append(ctx.apply_timed(A, x, y, 5))
        print(f"build {build}: " + " ".join(f"{t:.4f}" for t in ts) + " ms per product", flush=True)


if __name__ == "__main__":
    main()
