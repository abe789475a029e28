#!/usr/bin/env python3
"""tools/probe_twophase_va.py STEP_MB TRIES [BUILDS] — the product stream's physical pieces stay the same, the VIRTUAL address
they are mapped at moves in steps (SPMV_TP_VA_STEP_MB): does the address, not the memory, decide phase A's mode?"""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_VERBOSE"] = "1"


def main():
    step, tries = int(sys.argv[1]), int(sys.argv[2])
    builds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    os.environ.update(SPMV_TP_VA_STEP_MB=str(step), SPMV_TP_PLACEMENT_TRIES=str(tries))
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    held = []
    for b in range(builds):
        A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
        out = []
        for only in (1, 2, 0):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, 10) for _ in range(3)))
        A.set_param("twophase_only", 0)
        print(f"build {b}: A {out[0]:.4f}  B {out[1]:.4f}  both {out[2]:.4f} ms", flush=True)
        del A
        held.append(ctx.vector((1 + b) * (1 << 27)))


if __name__ == "__main__":
    main()
