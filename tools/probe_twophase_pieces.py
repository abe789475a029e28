#!/usr/bin/env python3
"""tools/probe_twophase_pieces.py [BUDGET_MB] [BUILDS] — the piece search of the two-phase layout (kernels_csr_twophase.hip:
tp_choose_pieces) on the C5 shard, several builds in one process with memory held between them: the time of every piece
timed (SPMV_TP_PLACEMENT_VERBOSE=1), what the search kept, and the product as built — against a build without the search."""
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
    budget = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    builds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    n, ncol, k = 10_000_000, 80_000_000, 32
    ctx = capi.Context(0)
    x, y = ctx.gen_vector(ncol, seed=1), ctx.vector(n)
    y.fill(0.0)
    held = []
    for b in range(builds):
        for mb in (budget, 0):
            os.environ["SPMV_TP_PLACEMENT_BUDGET_MB"] = str(mb)
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
            print(f"build {b}, budget {mb} MB: {t_build:.2f} s (generation + layout + search), handle holds {A.get_param('device_bytes') / 2**30:.2f} GB "
                  f"(free memory fell by {(free0 - free1) / 2**30:.2f} GB); pieces {A.get_param('twophase_pieces')}, timed {A.get_param('twophase_placements_timed')}, "
                  f"exchanged {A.get_param('twophase_pieces_exchanged')}, as built / kept {A.get_param('twophase_placement_spread') / 1000:.3f}; "
                  f"A {out[0]:.4f}  B {out[1]:.4f}  both {out[2]:.4f} ms", flush=True)
            del A
        held.append(ctx.vector((1 + b) * (1 << 27)))  # 1, 2, 3, ... GB held: the next builds start elsewhere


if __name__ == "__main__":
    main()
