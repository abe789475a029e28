"""What do fp64 device atomics on pinned, device-mapped HOST memory do on this box?

spmv_apply_host keeps y in device memory for kernels that add into y with atomics (abi.hip: adds_into_y_with_atomics) because
that is platform behaviour, not a HIP guarantee.  This probe forces the other route (SPMV_EXPERIMENTS=1 SPMV_HOST_Y_IN_PLACE=1:
y stays in the staging buffer whatever the kernel) and compares with the oracle-checked default, so that the header can SAY what
was seen.  Run: SPMV_EXPERIMENTS=1 SPMV_HOST_Y_IN_PLACE=1 python tools/probe_apply_host_atomics.py  (and once without, as control)
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
capi = pkg.capi
forced = os.environ.get("SPMV_EXPERIMENTS") == "1" and os.environ.get("SPMV_HOST_Y_IN_PLACE") == "1"
ctx = capi.Context(0)
rng = np.random.default_rng(41)
n = 60_000
lens = np.full(n, 2, np.int64)
lens[0] = n
rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
cc = np.empty(rp[-1], np.int32)
cc[:n] = np.arange(n)
cc[n::2] = 0
cc[n + 1::2] = np.arange(1, n)
cv = rng.uniform(-1, 1, rp[-1])
x = rng.uniform(0, 1, n)
rows = np.repeat(np.arange(n), lens)
ref = np.bincount(rows, weights=cv * x[cc], minlength=n)
scale = np.bincount(rows, weights=np.abs(cv) * x[cc], minlength=n)
print(f"y in place forced: {forced}; host_stores {ctx.get_param('host_stores')}")
A = ctx.csr(n, n, rp, cc, cv)
for kernel, mode, name in ((capi.CSR_SEGSCAN, 0, "segscan"), (capi.CSR_SPLIT, 1, "split, chunks"), (capi.CSR_VECTOR, 0, "row-parallel (no atomics)")):
    A.set_param("split_mode", mode)
    A.set_kernel(kernel)
    y = np.zeros(n)
    ctx.apply_host(A, x, y)
    t0 = time.perf_counter()
    for _ in range(200):
        ctx.apply_host(A, x, y)
    us = (time.perf_counter() - t0) / 200 * 1e6
    err = np.max(np.abs(y / 201 - ref) / np.maximum(scale, 1e-300))
    print(f"  {name:28s} atomics={A.get_param('adds_into_y_with_atomics')}  max |dy|/(|A||x|) = {err:.3e}  {us:7.1f} us per call")
O = ctx.coo(n, n, rows.astype(np.int32), cc, cv)
O.set_kernel(capi.CSR_VECTOR)
y = np.zeros(n)
ctx.apply_host(O, x, y)
print(f"  {'COO scan':28s} atomics={O.get_param('adds_into_y_with_atomics')}  max |dy|/(|A||x|) = {np.max(np.abs(y - ref) / np.maximum(scale, 1e-300)):.3e}")
