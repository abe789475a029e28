#!/usr/bin/env python3
"""tools/probe_small_host_calls.py - BASELINE configs[0]'s shape (10000 x 10000, 16 per row) through spmv_apply_host, per kernel:
microseconds per call against the resident product (spmv_apply + one wait per 200) and against launch + wait per call."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
capi, synth = pkg.capi, pkg.synth
ctx = capi.Context(0)
n, k = 10_000, 16
rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=1)
x = synth.vec_uniform(n, seed=1)
A = ctx.csr(n, n, rp, cc, cv)
dx, dy = ctx.vector_from(x), ctx.vector(n)
print(f"host_stores {ctx.get_param('host_stores')}; AUTO kept kernel {A.info.kernel} after {A.get_param('select_rounds')} rounds of {A.get_param('select_candidates')} candidates")
for kernel, name in ((0, "AUTO"), (1, "row-parallel"), (3, "scalar"), (4, "panel"), (2, "LDS window")):
    try:
        A.set_kernel(kernel)
    except capi.SpmvError as e:
        print(f"  {name}: {e}")
        continue
    y = np.zeros(n)
    for _ in range(20):
        ctx.apply_host(A, x, y)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            ctx.apply_host(A, x, y)
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    dy.fill(0.0)
    res = min(ctx.apply_timed(A, dx, dy, 200) for _ in range(3)) * 1e3
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.apply(A, dx, dy)
        ctx.sync()
    each = (time.perf_counter() - t0) / 200 * 1e6
    print(f"  {name:14s} kernel {A.info.kernel}: apply_host {best:6.2f} us per call ({2 * n * k / best / 1e3:5.1f} GFLOP/s)   resident, back to back {res:5.2f} us   "
          f"resident, launch + wait per call {each:6.2f} us")
