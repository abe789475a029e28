#!/usr/bin/env python3
"""tools/probe_twophase_classes.py [POOL] — which 1 GB pieces of device memory conflict as homes of the two-phase product
stream?  A shard whose stream takes two pieces; POOL pieces allocated one after the other; every piece is timed beside one
representative of every class found so far (same class = the pair is slow) and labelled.  Prints the labels in allocation
order: the pattern the allocator's memory comes in."""
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
    pool = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    n = 7_500_000
    ncol, k = 8 * n, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
    A.set_kernel(capi.CSR_TWOPHASE)
    assert A.get_param("twophase_pieces") == 2
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    A.set_param("twophase_pool_alloc", pool - 2)

    def t(a, b):
        A.set_param("twophase_pool_config", a | (b << 10))
        ctx.apply(A, x, y)
        return statistics.median(ctx.apply_timed(A, x, y, 4) for _ in range(2))

    # the two levels: piece 0 beside all others
    with0 = [t(0, b) for b in range(1, pool)]
    lo, hi = min(with0), max(with0)
    cut = 0.5 * (lo + hi)
    print(f"pairs with piece 0: fastest {lo:.4f} ms, slowest {hi:.4f} ms, cut {cut:.4f}")
    reps, label = [0], {0: 0}
    for p in range(1, pool):
        for c, r in enumerate(reps):
            if t(r, p) > cut:
                label[p] = c
                break
        else:
            label[p] = len(reps)
            reps.append(p)
    names = "ABCDEFGHIJKLMNOP"
    print("classes in allocation order:", " ".join(names[label[p]] for p in range(pool)))
    print("representatives:", reps, " pieces per class:", [sum(1 for p in label.values() if p == c) for c in range(len(reps))])
    # check: representatives beside each other
    for i, a in enumerate(reps):
        print(f"  {names[i]} beside the other representatives:", " ".join(f"{t(a, b):.3f}" for b in reps if b != a))


if __name__ == "__main__":
    main()
