#!/usr/bin/env python3
"""tools/probe_split_threshold.py - kernel SPLIT (kernels_csr_split.hip): which rows should count as long?

The default takes a sixteenth of the longest row (at least one chunk of 4096 entries).  On matrices whose row lengths fall off
gradually (R-MAT graphs, power laws) that leaves rows of thousands of entries to the short rows' kernel.  This probe times
the product over a range of thresholds on such matrices, each result checked against a host float64 product.

    python tools/probe_split_threshold.py [--out FILE]
"""
import argparse
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("sweep_structures", ROOT / "tools" / "sweep_structures.py")
ss = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ss)
capi = ss.capi


powerlaw = ss.powerlaw


CASES = {
    "rmat_18": lambda: ss.rmat(18, 16),
    "rmat_20": lambda: ss.rmat(20, 16),
    "rmat_22": lambda: ss.rmat(22, 16),
    "powerlaw_500k_cap_200k": lambda: powerlaw(500_000, 200_000),
    "powerlaw_1M_cap_500k": lambda: powerlaw(1_000_000, 500_000),
    "dense_rows_in_16M": lambda: ss.few_dense_rows(500_000, 32, 8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "probe_split_threshold.txt"))
    ap.add_argument("--cases", default="")
    ap.add_argument("--low", action="store_true", help="also thresholds of 32, 64 and 128 entries")
    args = ap.parse_args()
    ctx = capi.Context(0)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    with open(args.out, "a") as out:
        def say(s):
            print(s, flush=True)
            out.write(s + "\n")
        for name, build in CASES.items():
            if args.cases and not any(w in name for w in args.cases.split(",")):
                continue
            t0 = time.perf_counter()
            nrow, ncol, r, c, v = build()
            nnz = r.size
            x_h = np.random.default_rng(11).uniform(0.0, 1.0, ncol)
            ref = np.bincount(r, weights=v * x_h[c], minlength=nrow)
            scale = np.maximum(np.bincount(r, weights=np.abs(v) * x_h[c], minlength=nrow), 1e-300)
            ln = np.bincount(r, minlength=nrow)
            rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int32)
            x, y = ctx.vector_from(x_h), ctx.vector(nrow)
            say(f"{name}: {nrow} x {ncol}, {nnz} entries, rows mean {ln.mean():.1f} max {ln.max()}, "
                f"rows >= 1024 / 4096 / 16384: {(ln >= 1024).sum()} / {(ln >= 4096).sum()} / {(ln >= 16384).sum()} (built in {time.perf_counter() - t0:.1f}s)")
            t1 = time.perf_counter()
            A = ctx.csr(nrow, ncol, rp, c, v)
            ctx.sync()
            say(f"    handle created in {time.perf_counter() - t1:.2f} s (upload, analysis, every candidate's layout and timing)")
            say(f"    AUTO = kernel {A.info.kernel}" + (f" (threshold {A.get_param('split_row_threshold')}, {A.get_param('split_long_rows')} long rows, "
                f"inner kernel {A.get_param('split_inner_kernel')})" if A.info.kernel == 7 else "") + f": {ss.timed(ctx, A, x, y, 20):.4f} ms")
            for k, nm in ((4, "panel"), (6, "scan")):
                A.set_kernel(k)
                ss.check(ctx, A, x, y, ref, scale, f"{name} {nm}")
                say(f"    {nm}: {ss.timed(ctx, A, x, y, 20):.4f} ms")
            for mode, what in ((1, "chunks of 4096 entries"), (2, "virtual rows of 64 entries")):
                A.set_param("split_mode", mode)
                for T in ((32, 64, 128) if args.low else ()) + (256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 0):
                    A.set_param("split_row_threshold", T)
                    A.set_kernel(7)
                    ss.check(ctx, A, x, y, ref, scale, f"{name} split at {T} mode {mode}")
                    say(f"    split ({what}), rows >= {A.get_param('split_row_threshold'):>7d} long ({A.get_param('split_long_rows'):>6d} rows, "
                        f"{A.get_param('split_long_entries'):>10d} entries; short rows' kernel {A.get_param('split_inner_kernel')}, long rows' {A.get_param('split_long_kernel')}): "
                        f"{ss.timed(ctx, A, x, y, 20):.4f} ms" + ("   <- the default threshold" if T == 0 else ""))
            A.set_param("split_mode", 0)
            del A


if __name__ == "__main__":
    main()
