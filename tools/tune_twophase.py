#!/usr/bin/env python3
"""tools/tune_twophase.py — the two phases of the two-phase CSR product on the C5 shard shape (10M rows x 80M columns,
32 per row), timed separately and together, variants interleaved in one process (GPU box only).

    python tools/tune_twophase.py [--ncol 80000000] [--rounds 4] [--builds 1]

Switches of the engine (kernels_csr_twophase.hip): SPMV_TP_ONLY=1|2 one phase alone (per call; timing only), SPMV_TP_PAD=2|8|16
run padding and SPMV_TP_PLACEMENT_TRIES (at build); spmv_mat_set_param "twophase_unroll" picks the pairs per lane in flight
of the expand kernel.
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
os.environ["SPMV_EXPERIMENTS"] = "1"  # SPMV_TP_ONLY (one phase alone, wrong results) is honoured only with this


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--ncol", type=int, default=80_000_000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--builds", type=int, default=1, help="re-build the layout this many times (placement lottery)")
    ap.add_argument("--pads", type=lambda v: [int(t) for t in v.split(",")], default=[8], help="run padding per build, cycled: 2, 8 or 16 entries")
    ap.add_argument("--tries", type=int, default=12, help="placements of the product stream timed per build (1 = none)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, a.n, a.ncol, a.k, band=0, seed=1)
    x, y = ctx.gen_vector(a.ncol, seed=1), ctx.vector(a.n)
    y.fill(0.0)
    keys = ("SPMV_TP_ONLY", "SPMV_TP_ROTATE")
    for build in range(a.builds):
        pad = a.pads[build % len(a.pads)]
        os.environ["SPMV_TP_PAD"] = str(pad)
        os.environ["SPMV_TP_PLACEMENT_TRIES"] = str(a.tries)
        for cols in (10_000, 20_000):  # the first forces the re-build of the second: every stream is allocated again
            A.set_param("twophase_panel_cols", cols)
            A.set_kernel(capi.CSR_TWOPHASE)
        variants = [("A U3, every workgroup from its panel's start", dict(SPMV_TP_ONLY="1", SPMV_TP_ROTATE="0"), 3), ("A U3", dict(SPMV_TP_ONLY="1"), 3),
                    ("A U4", dict(SPMV_TP_ONLY="1"), 4), ("B", dict(SPMV_TP_ONLY="2"), 3), ("both U3", {}, 3)]
        res = {name: [] for name, _, _ in variants}
        for _ in range(a.rounds):
            for name, env, unroll in variants:
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(env)
                A.set_param("twophase_unroll", unroll)
                ctx.apply(A, x, y)
                res[name].append(ctx.apply_timed(A, x, y, a.reps))
        for k in keys:
            os.environ.pop(k, None)
        print(f"# build {build}: run padding {pad}, padded entries {A.get_param('twophase_padded')} ({A.get_param('twophase_padded') / A.info.nnz - 1:.2%} padding), "
              f"placements timed {A.get_param('twophase_placements_timed')}, slowest / kept {A.get_param('twophase_placement_spread') / 1000:.3f}")
        for name, ts in res.items():
            print(f"{name:32s} median {statistics.median(ts):.4f} ms   min {min(ts):.4f}", flush=True)


if __name__ == "__main__":
    main()
