#!/usr/bin/env python3
"""tools/probe_twophase_arena.py — the placement search of the two-phase layout (one arena of mapped pieces, windows timed
in turn: kernels_csr_twophase.hip, tp_choose_placement) run several times in one process with its per-window times printed
(SPMV_TP_PLACEMENT_VERBOSE=1), then the product as built.  Arguments: piece MB, step MB, extra MB, tries, builds."""
import os
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"
os.environ["SPMV_TP_PLACEMENT_VERBOSE"] = "1"


def main():
    piece, step, extra, tries, builds = (int(v) for v in (sys.argv[1:6] + ["256", "1024", "8192", "9", "4"][len(sys.argv) - 1:]))
    os.environ.update(SPMV_TP_ARENA_PIECE_MB=str(piece), SPMV_TP_ARENA_STEP_MB=str(step), SPMV_TP_ARENA_EXTRA_MB=str(extra),
                      SPMV_TP_PLACEMENT_TRIES=str(tries))
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    held = []
    for b in range(builds):
        free0, _ = ctx.mem_info()
        t = time.perf_counter()
        A = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=1)
        ctx.sync()
        t_build = time.perf_counter() - t
        free1, _ = ctx.mem_info()
        out = []
        for only in (1, 2, 0):
            A.set_param("twophase_only", only)
            ctx.apply(A, x, y)
            out.append(statistics.median(ctx.apply_timed(A, x, y, 10) for _ in range(3)))
        A.set_param("twophase_only", 0)
        print(f"build {b}: {t_build:.2f} s (generation + layout + search), handle holds {A.get_param('device_bytes') / 2**30:.2f} GB, free memory fell by "
              f"{(free0 - free1) / 2**30:.2f} GB; windows timed {A.get_param('twophase_placements_timed')}, slowest / kept "
              f"{A.get_param('twophase_placement_spread') / 1000:.3f}; A {out[0]:.4f}  B {out[1]:.4f}  both {out[2]:.4f} ms", flush=True)
        del A
        held.append(ctx.vector((1 + b) * (1 << 27)))  # 1, 2, 3, ... GB held: the next build starts elsewhere


if __name__ == "__main__":
    main()
