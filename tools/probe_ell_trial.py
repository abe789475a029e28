import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
capi = load_package().capi
ctx = capi.Context(0)
nb, b = 20000, 8
n = nb * b
i = np.arange(n)
K = b
ec = np.zeros(n * K, np.int32); ev = np.random.default_rng(1).uniform(-1, 1, n * K)
for s in range(K):
    ec[s * n:(s + 1) * n] = i // b * b + s
A = ctx.ell(n, n, K, n * K, ec, ev)
def rec(A): return {k: A.get_param("select_us_" + k) for k in ("vector", "variant1", "variant2", "panel")}
print("creation", rec(A), "variant", A.get_param("ell_variant"))
for t in range(3):
    A.set_kernel(0); print("again", rec(A), "variant", A.get_param("ell_variant"))
x = ctx.vector_from(np.random.default_rng(2).uniform(0, 1, n)); y = ctx.vector(n); y.fill(0.0)
z = ctx.vector(n); z.fill(0.0)
for lanes in (1, 2):
    A.set_kernel(1, lanes)
    for reps in (1, 4, 20, 50):
        ctx.apply(A, x, y)
        print("forced lanes", lanes, "reps", reps, "x real", round(min(ctx.apply_timed(A, x, y, reps) for _ in range(3)) * 1000, 2), "us",
              " x zeros", round(min(ctx.apply_timed(A, z, y, reps) for _ in range(3)) * 1000, 2), "us")
