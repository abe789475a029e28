#!/usr/bin/env python3
"""tools/probe_ell_placement.py - does WHERE an ELL handle's arrays lie decide C3's time inside one process?

BASELINE configs[2] (4M rows x 64 slots, circulant band) has measured 0.347 to 0.405 ms across runs and boxes with the same
kernel ("the spread is where its arrays lie").  Here: 8 handles of that matrix built one after the other and ALL
kept (each in other physical memory), every one timed 3 x 20 products, twice round.  Result (profiles/r05_probe_ell_placement.txt): they do differ, reproducibly per handle, 0.344 to 0.403 - but in RUNS of neighbours
(three slow then five fast; three fast then five slow; ...): re-homing a handle's 2 GB of values by timing four copies within the
two-phase search's 8 GB budget was built and removed again - all four copies usually lie in the same kind of memory as the
original, and the handles ended at 0.357-0.403 against 0.349-0.395 without it.
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    ctx = capi.Context(0)
    n, k = 4_000_000, 64
    x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
    handles = []
    for i in range(8):
        handles.append(ctx.gen_ell_banded(n, n, k, seed=1))
    ctx.sync()
    for rnd in range(2):
        line = []
        for A in handles:
            time.sleep(0.02)
            ctx.apply(A, x, y)
            ctx.apply(A, x, y)
            line.append(min(ctx.apply_timed(A, x, y, 20) for _ in range(3)))
        print(f"round {rnd}: " + "  ".join(f"{t:.4f}" for t in line) + f"   (ms per product; min {min(line):.4f}, max {max(line):.4f}, kernel {handles[0].info.kernel}, "
              f"diagonal slots {handles[0].get_param('ell_diagonal_slots')})", flush=True)


if __name__ == "__main__":
    main()
