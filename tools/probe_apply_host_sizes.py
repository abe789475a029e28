#!/usr/bin/env python3
"""tools/probe_apply_host_sizes.py - spmv_apply_host (host vectors in and out, the reference's call shape) over vector sizes: where
does the staged path (x by CPU stores into device memory, y updated in pinned memory, one launch) stop paying against the
asynchronous copies?  Run twice: SPMV_HOST_STAGED_MB=4 (the default limit) and =4096 (everything staged)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
capi, synth = pkg.capi, pkg.synth
ctx = capi.Context(0)
print(f"SPMV_HOST_STAGED_MB={os.environ.get('SPMV_HOST_STAGED_MB', '4 (default)')}")
for n in (10_000, 40_000, 160_000, 262_144, 600_000, 2_000_000, 8_000_000):
    k = 8
    rp, c, v = synth.csr_uniform(0, n, n, k, seed=1)
    A = ctx.csr(n, n, rp, c, v)
    x, y = synth.vec_uniform(n, seed=1), np.zeros(n)
    for _ in range(3):
        ctx.apply_host(A, x, y)
    reps = 200 if n <= 262_144 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.apply_host(A, x, y)
    us = (time.perf_counter() - t0) / reps * 1e6
    dx, dy = ctx.vector_from(x), ctx.vector(n)
    dy.fill(0.0)
    res = ctx.apply_timed(A, dx, dy, 20) * 1e3
    print(f"n = {n:>9d} (x + y = {16 * n / 1e6:7.2f} MB): {us:9.1f} us per call, {2 * n * k / us / 1e3:7.2f} GFLOP/s   (resident product {res:7.1f} us)", flush=True)
    del A, dx, dy
