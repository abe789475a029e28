#!/usr/bin/env python3
"""tools/probe_c4_split.py - BASELINE configs[3] (COO, N = 2M, power-law rows up to 4096, 115M entries) runs from its row-grouped
copy under the panel product: 0.27 ms = 0.66 of the bytes it moves, where the band-random C2 shape reaches 0.82 with the same
kernel.  Half of C4's entries sit in rows of hundreds to thousands of entries, and neighbouring lanes of the panel kernel then add
into the SAME LDS accumulator.  Does splitting the long rows off into virtual rows (SPMV_CSR_SPLIT, mode 2) at some threshold pay,
as it did on R-MAT graphs?  (AUTO does not time it here: the longest row holds 1/28000 of the entries.)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
ctx = capi.Context(0)
n = 2_000_000
G = ctx.gen_coo_powerlaw(n, n, 4096, seed=1)
x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
y.fill(0.0)


def timed(A):
    ctx.sync()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.01:
        ctx.apply_timed(A, x, y, 5)
    return min(ctx.apply_timed(A, x, y, 20) for _ in range(5))


print(f"COO handle, AUTO: kernel {G.info.kernel}, copy runs {G.get_param('rowgrouped_kernel')}: {timed(G):.4f} ms")
A = ctx.coo_to_csr(G)
print(f"CSR handle of the same entries, AUTO: kernel {A.info.kernel}: {timed(A):.4f} ms")
for mode in (2, 1):
    for thr in (4096, 2048, 1024, 512, 256, 128, 64):
        A.set_param("split_mode", mode)
        A.set_param("split_row_threshold", thr)
        t0 = time.perf_counter()
        A.set_kernel(capi.CSR_SPLIT)
        ctx.sync()
        build = time.perf_counter() - t0
        print(f"  split mode {mode} at {thr:5d}: {A.get_param('split_long_rows'):7d} long rows, {A.get_param('split_long_entries'):10d} entries, {A.get_param('split_virtual_rows'):8d} virtual rows, "
              f"inner kernel {A.get_param('split_inner_kernel')}, long kernel {A.get_param('split_long_kernel')}: {timed(A):.4f} ms   (built in {build:.2f} s)", flush=True)
A.set_param("split_row_threshold", 0)
A.set_param("split_mode", 0)
for rounds in (1, 2, 3, 4):
    A.set_param("panel_rounds", rounds)
    A.set_kernel(capi.CSR_PANEL)
    print(f"  panel layout cut for {rounds} round(s): {A.get_param('panel_groups')} groups: {timed(A):.4f} ms", flush=True)
