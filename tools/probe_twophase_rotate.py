#!/usr/bin/env python3
"""tools/probe_twophase_rotate.py — the expand phase's starting points ("twophase_rotate": how many distinct positions the
256 workgroups start their panels at) under a FAST and a SLOW placement of the product stream, built on purpose from a pool
of pieces (a shard whose stream takes two 1 GB pieces: pieces of different classes / of the same class, found by timing
pairs with piece 0)."""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_BUDGET_MB"] = "8192"


def main():
    n, pool = 7_500_000, 14
    ncol, k = 8 * n, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    assert A.get_param("twophase_pieces") == 2
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    A.set_param("twophase_pool_alloc", pool - 2)

    def t(cfg, only=0, reps=6):
        A.set_param("twophase_pool_config", cfg[0] | (cfg[1] << 10))
        A.set_param("twophase_only", only)
        ctx.apply(A, x, y)
        v = statistics.median(ctx.apply_timed(A, x, y, reps) for _ in range(3))
        A.set_param("twophase_only", 0)
        return v

    with0 = [(t((0, b)), b) for b in range(1, pool)]
    fast, slow = min(with0), max(with0)
    print("piece 0 with every other piece:", " ".join(f"{b}:{v:.3f}" for v, b in with0))
    print(f"fast placement (0, {fast[1]}) {fast[0]:.4f} ms, slow placement (0, {slow[1]}) {slow[0]:.4f} ms")
    print(f"{'starting points':>16} {'fast: A':>9} {'both':>9} {'slow: A':>9} {'both':>9}")
    for rot in (0, 2, 4, 8, 16, 32, 64, 128, 256):
        A.set_param("twophase_rotate", rot)
        row = [t((0, fast[1]), 1), t((0, fast[1]), 0), t((0, slow[1]), 1), t((0, slow[1]), 0)]
        print(f"{rot:>16d} " + " ".join(f"{v:9.4f}" for v in row), flush=True)


if __name__ == "__main__":
    main()
