#!/usr/bin/env python3
"""tools/probe_panel_placement.py — C2 through the panel kernel, its layout built several times in one process (two row
group sizes alternate, so every build frees and allocates the layout again): how much of the run-to-run spread of the
headline number is the memory the layout happens to get?"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi


def main():
    n, k = 10_000_000, 32
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, n, n, k, band=0, seed=1)
    x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
    y.fill(0.0)
    for k_, v in (("panel_aos", 3), ("panel_unroll", 8), ("panel_pipe", 2), ("panel_sync", 3), ("panel_trial", 0)):
        A.set_param(k_, v)
    for build in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
        for rows in (19_000, 0):  # the first forces the re-build of the second (0 = the default: 19532)
            A.set_param("panel_rows", rows)
            A.set_kernel(capi.CSR_PANEL)
        ts = []
        for _ in range(3):
            ctx.apply(A, x, y)
            ts.append(ctx.apply_timed(A, x, y, 10))
        print(f"build {build}: rows {A.get_param('panel_rows')}: " + " ".join(f"{t:.4f}" for t in ts) + " ms per product", flush=True)


if __name__ == "__main__":
    main()
