#!/usr/bin/env python3
"""tools/probe_panel_width_wide_x.py - the panel width (columns of x per panel) on shards whose x is 2x / 4x their rows
(what a rank holds at N = 2 / N = 4 of BASELINE's weak scaling: 10M rows x 32 over 20M / 40M columns).  The default of 131072
columns (1 MB of x) was chosen on C2 (x = 80 MB)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def timed(ctx, A, x, y, reps=20):
    ctx.sync()
    time.sleep(0.02)
    ctx.apply(A, x, y)
    ctx.apply(A, x, y)
    return min(ctx.apply_timed(A, x, y, reps) for _ in range(3))


def main():
    ctx = capi.Context(0)
    n, k = 10_000_000, 32
    for ncol in (20_000_000, 40_000_000):
        A = ctx.gen_csr_uniform(0, n, ncol, k, seed=1)
        x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
        print(f"{n} rows x {k} over {ncol} columns: AUTO = kernel {A.info.kernel}: {timed(ctx, A, x, y):.4f} ms", flush=True)
        for w in (32768, 65536, 131072, 262144, 524288):
            A.set_param("panel_width", w)
            A.set_kernel(capi.CSR_PANEL)
            print(f"    panel, {w:>8d} columns per panel: {timed(ctx, A, x, y):.4f} ms  (unroll {A.get_param('panel_unroll')}, order {A.get_param('panel_pipe')}, "
                  f"barrier {A.get_param('panel_sync')}, groups {A.get_param('panel_groups')})", flush=True)
        del A, x, y


if __name__ == "__main__":
    main()
