#!/usr/bin/env python3
"""tools/probe_twophase_vm.py — does HOW the product stream of the two-phase layout is allocated decide its mode?
Round 3 found phase A at 1.17-1.20 or 1.32-1.35 ms by WHERE hipMalloc put the stream (profiles/r03_probe_twophase_*).  This
moves the stream again and again under three allocators ("twophase_alloc_mode": 0 hipMalloc; 1 physical memory created in
one piece and mapped at a 1 GB-aligned virtual address; 2 the same in pieces of 1 GB), with a spacer held between some moves,
and prints phase A / phase B alone after every move."""
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
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)

    def phase(only):
        A.set_param("twophase_only", only)
        ctx.apply(A, x, y)
        t = statistics.median(ctx.apply_timed(A, x, y, 5) for _ in range(3))
        A.set_param("twophase_only", 0)
        return t

    print(f"as built (hipMalloc): A {phase(1):.4f}  B {phase(2):.4f} ms", flush=True)
    held = []
    for mode in (1, 2, 0, 1, 2, 0):
        A.set_param("twophase_alloc_mode", mode)
        for spacer_gb in (0, 0, 0, 3, 0, 5, 0, 0):
            if spacer_gb:
                held.append(ctx.vector(spacer_gb * (1 << 27)))  # doubles: 2^27 * 8 B = 1 GB
            A.set_param("twophase_realloc", 1)
            free, _total = ctx.mem_info()
            print(f"mode {mode}, spacer {spacer_gb} GB held before this move (free {free / 2**30:.0f} GB): A {phase(1):.4f}  B {phase(2):.4f} ms", flush=True)
        held.clear()


if __name__ == "__main__":
    main()
