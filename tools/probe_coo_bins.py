#!/usr/bin/env python3
"""tools/probe_coo_bins.py [--ncol N] [--reps R] — the COO segmented scan of C4 (2M rows, power-law up to 4096) over the
entries as stored and over the copy in 8 / 16 / 32 / 64 column bins (coo_column_bins = 0, 1, 2, 4, 8 per XCD), and, as
the scan's own ceiling, the same rows with columns drawn from 100 000 (x = 0.8 MB: every gather hits L2)."""
import argparse
import pathlib
import statistics
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ncol", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", type=int, default=-1, help="one setting of coo_column_bins only (for a counter pass)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    n = 2_000_000
    for ncol in (a.ncol, 100_000) if a.only < 0 else (a.ncol,):
        A = ctx.gen_coo_powerlaw(n, ncol, 4096, seed=1)
        nnz = A.info.nnz
        x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
        y.fill(0.0)
        A.set_kernel(capi.CSR_VECTOR)
        alg = 16 * nnz + 8 * ncol + 16 * n
        print(f"COO n={n} ncol={ncol} nnz={nnz}: 16 B per entry = {alg / 1e9:.3f} GB per product")
        for per_xcd in ((0, 1, 2, 4, 8) if a.only < 0 else (a.only,)):
            A.set_param("coo_column_bins", per_xcd)
            ts = []
            for _ in range(a.rounds):
                ctx.apply(A, x, y)
                ts.append(ctx.apply_timed(A, x, y, a.reps))
            mn = min(ts)
            print(f"  bins per XCD {per_xcd}: median {statistics.median(ts):.4f} ms, min {mn:.4f} ms = {alg / mn / 1e6:.0f} GB/s = {alg / mn / 1e6 / 80:.1f} % of 8 TB/s"
                  f"  (padded entries {A.get_param('coo_bins_padded')})")


if __name__ == "__main__":
    main()
