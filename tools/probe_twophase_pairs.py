#!/usr/bin/env python3
"""tools/probe_twophase_pairs.py [ROWS] [POOL] — is the placement effect of the two-phase product stream a matter of PAIRS of
pieces?  A shard whose stream fits two 1 GB pieces (7.5M rows x 32, x of 8 x as many columns), a pool of POOL pieces, the
product timed with every ordered pair (a, b) under the stream: the matrix of times shows which pieces go together."""
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_BUDGET_MB"] = "8192"  # whole pieces (the search itself is not the subject)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 7_500_000
    pool = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ncol, k = 8 * n, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    need = A.get_param("twophase_pieces")
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    A.set_param("twophase_pool_alloc", pool - need)
    print(f"{n} rows x {k}, {ncol} columns: the stream takes {need} piece(s); pool of {pool}", flush=True)

    def t(cfg):
        code = sum(p << (10 * i) for i, p in enumerate(cfg))
        A.set_param("twophase_pool_config", code)
        ctx.apply(A, x, y)
        return statistics.median(ctx.apply_timed(A, x, y, 4) for _ in range(3))

    if need == 2:
        print("ms per product, row = piece under the first gigabyte, column = piece under the second:")
        print("      " + " ".join(f"{b:6d}" for b in range(pool)))
        for a in range(pool):
            row = [t((a, b)) if a != b else float("nan") for b in range(pool)]
            print(f"{a:4d}  " + " ".join(f"{v:6.3f}" for v in row), flush=True)
    else:
        import itertools
        import random
        random.seed(1)
        combos = list(itertools.permutations(range(pool), need))
        random.shuffle(combos)
        for cfg in combos[:150]:
            print(cfg, f"{t(cfg):.4f}", flush=True)


if __name__ == "__main__":
    main()
