#!/usr/bin/env python3
"""tools/probe_lds_conflicts.py - do the long rows of an R-MAT graph make lanes of the panel kernel add into the same LDS word?

Run under the profiler (counters in a run of their own):
    cd /tmp && rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d OUT -- python3 tools/probe_lds_conflicts.py
    python3 tools/pmc_summary.py OUT csr_panel_pp --runs
The script runs, in this order, 3 products each: R-MAT scale 22 under the panel kernel; the same handle with every row of 256
entries and more split off into virtual rows (two panel launches per product: the short rows, the virtual rows); a power law
with uniform columns (500000 rows of min(200000, 8/u) entries) under the panel kernel as the control.
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
spec = importlib.util.spec_from_file_location("sweep_structures", ROOT / "tools" / "sweep_structures.py")
ss = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ss)
capi = ss.capi


def main():
    ctx = capi.Context(0)
    for name, build in (("rmat_22", lambda: ss.rmat(22, 16)), ("powerlaw_500k_cap_200k", lambda: ss.powerlaw(500_000, 200_000))):
        nrow, ncol, r, c, v = build()
        ln = np.bincount(r, minlength=nrow)
        rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int32)
        x, y = ctx.vector_from(np.random.default_rng(11).uniform(0.0, 1.0, ncol)), ctx.vector(nrow)
        import os
        os.environ["SPMV_PANEL_TRIAL"] = "0"  # no timing launches: every panel dispatch in the trace is a product
        A = ctx.csr(nrow, ncol, rp, c, v)
        A.set_kernel(4)
        for _ in range(3):
            ctx.apply(A, x, y)
        ctx.sync()
        print(f"{name}: {r.size} entries; 3 products under the panel kernel", flush=True)
        if name == "rmat_22":
            A.set_param("split_row_threshold", 256)
            A.set_param("split_mode", 2)
            A.set_kernel(7)
            for _ in range(3):
                ctx.apply(A, x, y)
            ctx.sync()
            print(f"{name}: 3 products with {A.get_param('split_long_rows')} rows split off ({A.get_param('split_long_entries')} entries in "
                  f"{A.get_param('split_virtual_rows')} virtual rows; kernels {A.get_param('split_inner_kernel')} / {A.get_param('split_long_kernel')})", flush=True)
        del A, x, y


if __name__ == "__main__":
    main()
