#!/usr/bin/env python3
"""tools/probe_c3_layouts.py - BASELINE configs[2] (4M rows x 64 slots, circulant band) in one process, handles interleaved:
the ELL handle as AUTO leaves it (diagonal slots, column-major values), the same with the values in tiles of 512 rows
("ell_tiled_values"), and the engine's DIA handle of the same band (row-major values, the DIA kernel).  Does the DIA order
beat the ELL kernels on the SAME box, in memory allocated at the same moment?  """
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def timed(ctx, A, x, y):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.01:
        ctx.apply_timed(A, x, y, 5)
    return min(ctx.apply_timed(A, x, y, 20) for _ in range(4))


def main():
    ctx = capi.Context(0)
    n, k = 4_000_000, 64
    x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
    sets = []
    for i in range(4):
        E = ctx.gen_ell_banded(n, n, k, seed=1)
        T = ctx.gen_ell_banded(n, n, k, seed=1)
        T.set_param("ell_tiled_values", 1)
        D = ctx.gen_dia_banded(n, k, seed=1)
        extra = None
        try:
            R = ctx.gen_ell_banded(n, n, k, seed=1)
            R.set_param("ell_dia_order", 1)
            extra = R
        except Exception:  # (an engine without the DIA-order copy)
            pass
        sets.append((E, T, D, extra))
    ctx.sync()
    moved = 8.0 * n * k + 8.0 * n + 16.0 * n
    for rnd in range(2):
        for i, (E, T, D, R) in enumerate(sets):
            te, tt, td = timed(ctx, E, x, y), timed(ctx, T, x, y), timed(ctx, D, x, y)
            tr = timed(ctx, R, x, y) if R is not None else float("nan")
            print(f"round {rnd} set {i}: ELL {te:.4f}  ELL tiled {tt:.4f}  DIA {td:.4f}  ELL + DIA-order copy {tr:.4f} ms   "
                  f"({moved / te / 1e6:.0f} / {moved / tt / 1e6:.0f} / {moved / td / 1e6:.0f} / {moved / tr / 1e6:.0f} GB/s of the 2.14 GB a product moves)", flush=True)


if __name__ == "__main__":
    main()
