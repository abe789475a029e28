import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
capi = load_package().capi
ctx = capi.Context(0)
n = 1_000_000
col = np.zeros(n, np.int32); val = np.random.default_rng(1).uniform(-1, 1, n)
x, y = ctx.vector_from(np.array([0.5])), ctx.vector(n)
def timed(A):
    ctx.sync(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.003: ctx.apply_timed(A, x, y, 50)
    return min(ctx.apply_timed(A, x, y, 50) for _ in range(8)) * 1e3
A = ctx.ell(n, 1, 1, n, col, val)
def state(): return f"kernel {A.info.kernel} lanes {A.info.lanes_per_row} variant {A.get_param('ell_variant')} diag {A.get_param('ell_diagonal_slots')}"
print("AUTO after creation", state(), f"{timed(A):.2f} us")
for rep in range(2):
    A.set_kernel(1, 2); print("forced (1,2)      ", state(), f"{timed(A):.2f} us")
    A.set_kernel(1, 1); print("forced (1,1)      ", state(), f"{timed(A):.2f} us")
    A.set_kernel(1, 2); print("forced (1,2)      ", state(), f"{timed(A):.2f} us")
    A.set_kernel(0);    print("AUTO again        ", state(), f"{timed(A):.2f} us")
    A.set_kernel(1, 0); print("forced (1,0)      ", state(), f"{timed(A):.2f} us")
B = ctx.ell(n, 1, 1, n, col, val)
B.set_kernel(1, 2)
print("second handle forced (1,2)", f"{timed(B):.2f} us")
