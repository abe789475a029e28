#!/usr/bin/env python3
"""tools/sweep_sizes.py — the AUTO kernel choice against the forced row-parallel kernel over a grid of CSR shapes
(GPU box only).  A sanity map of the selection policy (csr_choose_kernel): AUTO should never lose by much."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402

capi = load_package().capi
NAMES = {1: "vector", 2: "ldswin", 3: "scalar", 4: "panel", 5: "twophase", 6: "segscan", 7: "split", 8: "ell"}


def timed(ctx, A, x, y, reps):
    ctx.apply(A, x, y)
    return min(ctx.apply_timed(A, x, y, reps) for _ in range(3))


def main():
    ctx = capi.Context(0)
    print(f"{'rows':>9s} {'k':>3s} {'band':>6s} | {'auto':>7s} {'ms':>8s} {'GFLOP/s':>8s} {'%8TB/s':>7s} | {'vector ms':>9s} {'auto/vector':>11s}")
    for n in (20_000, 100_000, 400_000, 1_000_000, 4_000_000, 10_000_000, 20_000_000):
        for k in (4, 16, 32, 64):
            if n * k > 700_000_000:
                continue
            for band in (0, 4096):
                if band and band >= n:
                    continue
                A = ctx.gen_csr_uniform(0, n, n, k, band=band, seed=3)
                x, y = ctx.gen_vector(n, seed=3), ctx.vector(n)
                y.fill(0.0)
                reps = 20 if n * k < 50_000_000 else 5
                kern = NAMES.get(int(A.info.kernel), str(A.info.kernel))
                t_auto = timed(ctx, A, x, y, reps)
                A.set_kernel(capi.CSR_VECTOR)
                t_vec = timed(ctx, A, x, y, reps)
                nnz = n * k
                gb = algorithmic_bytes("csr", n, n, nnz) / t_auto / 1e6
                print(f"{n:9d} {k:3d} {band:6d} | {kern:>7s} {t_auto:8.4f} {2 * nnz / t_auto / 1e6:8.1f} {gb / 80:7.2f} | {t_vec:9.4f} {t_vec / t_auto:10.2f}x",
                      flush=True)
                del A, x, y


if __name__ == "__main__":
    main()
