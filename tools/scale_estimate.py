#!/usr/bin/env python3
"""tools/scale_estimate.py — what the weak-scaling curve of bench.py should look like, measured on ONE GPU: for N = 1, 2, 4, 8 the
shard rank N-1 would hold (10M rows of the 10M*N x 10M*N matrix, 32 per row) is generated, laid out by the automatic policy and
timed (x resident, as in the headline loop).  With the exchange outside the timed loop the ranks are independent, so the N-GPU
aggregate is N shards per shard time.  Uniform-random columns (BASELINE configs[1] / [4]) and the band-random variant."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
NAMES = {1: "row-parallel", 2: "LDS window", 3: "scalar", 4: "panel", 5: "two-phase", 6: "segmented scan", 7: "long-row split", 8: "ELL copy"}


def main():
    n, k = 10_000_000, 32
    ctx = capi.Context(0)
    for band in (0, 65536):
        print(f"# {'uniform-random columns' if band == 0 else f'columns random in a band of {band}'}: N, kernel, ms per product, aggregate GFLOP/s, speed-up over N = 1")
        base = None
        for world in (1, 2, 4, 8):
            ncol = n * world
            A = ctx.gen_csr_uniform((world - 1) * n, world * n, ncol, k, band=band, seed=1)
            if A.info.kernel in (4, 5):
                A.set_param("panel_keep_csr", 0)
            x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
            y.fill(0.0)
            for _ in range(5):
                ctx.apply(A, x, y)
            ms = min(ctx.apply_timed(A, x, y, 50) for _ in range(3))
            gf = 2.0 * n * k * world / ms / 1e6
            base = base or gf
            print(f"N={world}  {NAMES.get(A.info.kernel, A.info.kernel):10s}  {ms:.4f} ms  {gf:8.1f} GFLOP/s  {gf / base:.2f}x", flush=True)
            del A, x, y


if __name__ == "__main__":
    main()
