#!/usr/bin/env python3
"""tools/fuzz_medium.py - random matrices at the sizes where handles TIME their candidates (64K .. ~10M entries).

The GPU fuzz tests (tests/test_gpu_fuzz.py) use small matrices, below most of the selection's thresholds.  Here every case
draws a shape (square, tall, wide, one column ...), a row-length law (constant, uniform, power law with a cap, a few dense rows,
stretches of empty rows), a column law (uniform, a band, hub columns, contiguous runs), an entry ORDER for the COO handle
(row-sorted, column-sorted as .mtx files come, shuffled with duplicates), and runs the product through
  * a CSR handle under AUTO and under every kernel that can be forced (SEGSCAN; SPLIT in both modes at a random threshold),
  * a COO, a CSC and (where the padding stays below 4x) an ELL handle under AUTO,
each against a host float64 product scaled by |A||x| (1e-10), y starting from a random vector (y += A x).

    python tools/fuzz_medium.py [--cases 40] [--seed 1] [--out FILE]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
NAMES = {1: "vector", 2: "ldswin", 3: "scalar", 4: "panel", 5: "twophase", 6: "segscan", 7: "split", 8: "ell"}


def draw(rng):
    shape = rng.choice(["square", "tall", "wide", "narrow"], p=[0.5, 0.2, 0.2, 0.1])
    nrow = int(rng.integers(2_000, 1_500_000))
    ncol = {"square": nrow, "tall": max(1, nrow // int(rng.integers(4, 400))), "wide": nrow * int(rng.integers(2, 12)),
            "narrow": int(rng.integers(1, 40))}[shape]
    law = rng.choice(["constant", "uniform", "powerlaw", "dense_rows", "empty_stretches"])
    target = int(rng.integers(80_000, 6_000_000))
    mean = max(1, target // nrow)
    if law == "constant":
        ln = np.full(nrow, mean, np.int64)
    elif law == "uniform":
        ln = rng.integers(0, 2 * mean + 1, nrow)
    elif law == "powerlaw":
        cap = int(rng.choice([64, 4096, 100_000, 1_000_000]))
        ln = np.minimum(cap, (max(1, mean // 6) / np.maximum(rng.random(nrow), 1e-9)).astype(np.int64))
    elif law == "dense_rows":
        ln = np.full(nrow, mean, np.int64)
        ln[rng.choice(nrow, int(rng.integers(1, 6)), replace=False)] = min(ncol, int(rng.integers(20_000, 700_000)))
    else:
        ln = np.full(nrow, 2 * mean, np.int64)
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, nrow))
            ln[a:a + int(rng.integers(1, max(2, nrow // 3)))] = 0
    ln = np.minimum(ln, ncol * 4)  # (duplicates allowed, but keep it bounded)
    while ln.sum() > 12_000_000:
        ln = ln // 2
    nnz = int(ln.sum())
    r = np.repeat(np.arange(nrow, dtype=np.int64), ln)
    claw = rng.choice(["uniform", "band", "hubs", "runs", "diagonals"])
    if claw == "diagonals" and ncol >= 64 and int(ln.max()) <= 512:
        # slot s of every row holds column i' + off[s] (i' = the row scaled to the columns): a stencil / band as ELL stores it;
        # wrapped at the edges, and a few rows with arbitrary columns (round 6: diagonal slots, the DIA-order copy)
        offs = np.sort(rng.choice(np.arange(-min(ncol // 2, 3000), min(ncol // 2, 3000)), size=int(ln.max()), replace=False))
        rp0 = np.concatenate(([0], np.cumsum(ln)))
        slot = np.arange(nnz) - np.repeat(rp0[:-1], ln)
        c = (r * ncol // nrow + offs[slot]) % ncol
        odd = rng.random(nnz) < 0.0005
        c[odd] = rng.integers(0, ncol, int(odd.sum()))
    elif claw == "uniform" or ncol < 64:
        c = rng.integers(0, ncol, nnz)
    elif claw == "band":
        w = int(rng.integers(8, max(9, ncol // 10)))
        c = np.clip((r * ncol // nrow) + rng.integers(-w, w + 1, nnz), 0, ncol - 1)
    elif claw == "hubs":
        c = np.where(rng.random(nnz) < 0.3, rng.integers(0, min(ncol, 64), nnz), rng.integers(0, ncol, nnz))
    else:
        start = rng.integers(0, ncol, nrow)
        rp = np.concatenate(([0], np.cumsum(ln)))
        c = (np.repeat(start, ln) + (np.arange(nnz) - np.repeat(rp[:-1], ln))) % ncol
    v = rng.uniform(-1.0, 1.0, nnz)
    return f"{shape} {nrow} x {ncol}, rows {law}, columns {claw}", nrow, ncol, r, c.astype(np.int64), v, ln


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "fuzz_medium.txt"))
    args = ap.parse_args()
    ctx = capi.Context(0)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    worst, nprod, t0 = 0.0, 0, time.perf_counter()
    with open(args.out, "a") as out:
        def say(s):
            print(s, flush=True)
            out.write(s + "\n")
        for case in range(args.cases):
            rng = np.random.default_rng(args.seed * 1000 + case)
            what, nrow, ncol, r, c, v, ln = draw(rng)
            nnz = r.size
            xh, y0 = rng.uniform(-1.0, 1.0, ncol), rng.uniform(-1.0, 1.0, nrow)
            ref = y0 + np.bincount(r, weights=v * xh[c], minlength=nrow)
            scale = np.maximum(np.bincount(r, weights=np.abs(v) * np.abs(xh[c]), minlength=nrow) + np.abs(y0), 1e-300)
            rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int32)
            x = ctx.vector_from(xh)
            done = []

            def product(A, label):
                nonlocal worst, nprod
                y = ctx.vector_from(y0)
                ctx.apply(A, x, y)
                ctx.sync()
                err = float(np.max(np.abs(y.download() - ref) / scale))
                worst, nprod = max(worst, err), nprod + 1
                if not err <= 1e-10:
                    raise AssertionError(f"case {case} ({what}, {nnz} entries) {label}: |dy| / (|A||x| + |y0|) = {err:.3e}")
                done.append(label)

            def from_plan(A, make, label):
                # round 6: another handle of the same matrix built from A's plan runs A's kernel (and its copies' kernels) - no trial
                plan = A.get_plan()
                B = make()
                B.set_plan(plan)
                if B.get_plan() != plan or int(B.info.kernel) != int(A.info.kernel) or B.get_param("select_candidates") != 0:
                    raise AssertionError(f"case {case} ({what}) {label}: the handle built from the plan differs from the one the plan came from")
                product(B, label + " from its plan")

            A = ctx.csr(nrow, ncol, rp, c.astype(np.int32), v)
            product(A, f"csr auto={NAMES.get(int(A.info.kernel), A.info.kernel)}")
            from_plan(A, lambda: ctx.csr(nrow, ncol, rp, c.astype(np.int32), v), "csr")
            for k in (1, 3, 4, 6):
                A.set_kernel(k)
                product(A, f"csr {NAMES[k]}")
            if nrow * int(ln.max()) <= 8 * max(nnz, 1) and nrow * int(ln.max()) < 60_000_000 and int(ln.min()) >= 1:  # (refused with an empty row)
                A.set_kernel(8)
                product(A, f"csr ell copy ({'diagonal slots' if A.get_param('ell_copy_diagonal_slots') else 'columns'}, variant {A.get_param('ell_copy_variant')})")
            for mode in (1, 2):
                A.set_param("split_mode", mode)
                A.set_param("split_row_threshold", int(rng.choice([0, 1, 3, 64, 1000, 50_000])))
                A.set_kernel(7)
                product(A, f"csr split mode {mode} at {A.get_param('split_row_threshold')} ({A.get_param('split_long_rows')} long rows)")
            del A
            order = rng.choice(["rows", "columns", "shuffled"])
            perm = {"rows": np.arange(nnz), "columns": np.lexsort((r, c)), "shuffled": rng.permutation(nnz)}[order]
            A = ctx.coo(nrow, ncol, r[perm].astype(np.int32), c[perm].astype(np.int32), v[perm])
            product(A, f"coo ({order}) auto={'copy:' + NAMES.get(A.get_param('rowgrouped_kernel'), '?') if int(A.info.kernel) == 4 else 'scan'}")
            from_plan(A, lambda: ctx.coo(nrow, ncol, r[perm].astype(np.int32), c[perm].astype(np.int32), v[perm]), "coo")
            del A
            cs = np.lexsort((r, c))
            cp = np.concatenate(([0], np.cumsum(np.bincount(c, minlength=ncol)))).astype(np.int32)
            A = ctx.csc(nrow, ncol, cp, r[cs].astype(np.int32), v[cs])
            product(A, f"csc auto={'copy:' + NAMES.get(A.get_param('rowgrouped_kernel'), '?') if int(A.info.kernel) == 4 else 'scatter'}")
            from_plan(A, lambda: ctx.csc(nrow, ncol, cp, r[cs].astype(np.int32), v[cs]), "csc")
            del A
            K = int(ln.max()) if nrow else 0
            if 0 < K and nrow * K <= 4 * max(nnz, 1) and nrow * K < 40_000_000:
                ec, ev = np.zeros(nrow * K, np.int32), np.zeros(nrow * K)
                pos = np.arange(nnz) - np.repeat(rp[:-1].astype(np.int64), ln)
                ec[pos * nrow + r] = c
                ev[pos * nrow + r] = v
                A = ctx.ell(nrow, ncol, K, nnz, ec, ev)
                product(A, f"ell K={K} auto={'copy:' + NAMES.get(A.get_param('rowgrouped_kernel'), '?') if int(A.info.kernel) == 4 else 'own variant ' + str(A.get_param('ell_variant'))}")
                from_plan(A, lambda: ctx.ell(nrow, ncol, K, nnz, ec, ev), "ell")
                if A.get_param("ell_diagonal_slots"):
                    # the DIA-order copy against two rows per lane over the column-major values: the same slot order, the same bits
                    def bits(M):
                        yy = ctx.vector_from(y0)
                        ctx.apply(M, x, yy)
                        ctx.sync()
                        return yy.download()
                    A.set_kernel(1, 2)
                    own = bits(A)
                    A.set_param("ell_dia_order", 1)
                    if not np.array_equal(bits(A), own):
                        raise AssertionError(f"case {case} ({what}): the DIA-order copy of an ELL handle differs from the ELL kernel in some bit")
                    product(A, f"ell DIA order ({A.get_param('ell_non_conforming_rows')} rows off their diagonals)")
                del A
            say(f"case {case:3d}: {what}; {nnz} entries, longest row {int(ln.max())}: " + "; ".join(done))
        say(f"# {args.cases} cases, {nprod} products, all within 1e-10 (worst {worst:.2e}); {time.perf_counter() - t0:.0f} s")


if __name__ == "__main__":
    main()
