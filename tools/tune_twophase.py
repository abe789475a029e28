#!/usr/bin/env python3
"""tools/tune_twophase.py — the two phases of the two-phase CSR product on the C5 shard shape (10M rows x 80M columns,
32 per row), timed separately and together, variants interleaved in one process (GPU box only).

    python tools/tune_twophase.py [--ncol 80000000] [--rounds 4] [--builds 1]

Switches of the engine (kernels_csr_twophase.hip), all per handle through spmv_mat_set_param: "twophase_only" 1|2 one phase
alone (timing only: needs SPMV_EXPERIMENTS=1), "twophase_rotate", "twophase_placement_budget_mb" (the piece search's memory
budget; 0 = no search); SPMV_TP_PAD=2|8|16 run padding is read when the layout is built.
"""
import argparse
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
os.environ["SPMV_EXPERIMENTS"] = "1"  # "twophase_only" (one phase alone, wrong results) exists only with this


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--ncol", type=int, default=80_000_000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--builds", type=int, default=1, help="re-build the layout this many times (placement lottery)")
    ap.add_argument("--pads", type=lambda v: [int(t) for t in v.split(",")], default=[8], help="run padding per build, cycled: 2, 8 or 16 entries")
    ap.add_argument("--budget-mb", type=int, default=8192, help="memory the piece search may hold beyond the stream (0 = no search)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, a.n, a.ncol, a.k, band=0, seed=1)
    x, y = ctx.gen_vector(a.ncol, seed=1), ctx.vector(a.n)
    y.fill(0.0)
    for build in range(a.builds):
        pad = a.pads[build % len(a.pads)]
        os.environ["SPMV_TP_PAD"] = str(pad)
        A.set_param("twophase_placement_budget_mb", a.budget_mb)
        for cols in (10_000, 20_000):  # the first forces the re-build of the second: every stream is allocated again
            A.set_param("twophase_panel_cols", cols)
            A.set_kernel(capi.CSR_TWOPHASE)
        # name, phase alone (0 = both), rotate
        variants = [("A, every workgroup from its panel's start", 1, 0), ("A", 1, 256), ("B", 2, 256), ("both", 0, 256)]
        res = {v[0]: [] for v in variants}
        for _ in range(a.rounds):
            for name, only, rotate in variants:
                A.set_param("twophase_only", only)
                A.set_param("twophase_rotate", rotate)
                ctx.apply(A, x, y)
                res[name].append(ctx.apply_timed(A, x, y, a.reps))
        A.set_param("twophase_only", 0)
        print(f"# build {build}: run padding {pad}, padded entries {A.get_param('twophase_padded')} ({A.get_param('twophase_padded') / A.info.nnz - 1:.2%} padding), "
              f"pieces {A.get_param('twophase_pieces')} (timed {A.get_param('twophase_placements_timed')}, exchanged {A.get_param('twophase_pieces_exchanged')}, "
              f"as built / kept {A.get_param('twophase_placement_spread') / 1000:.3f})")
        for name, ts in res.items():
            print(f"{name:32s} median {statistics.median(ts):.4f} ms   min {min(ts):.4f}", flush=True)


if __name__ == "__main__":
    main()
