import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
capi = load_package().capi
ctx = capi.Context(0)
def timed(A, x, y):
    ctx.sync(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.005: ctx.apply_timed(A, x, y, 10)
    return min(ctx.apply_timed(A, x, y, 20) for _ in range(6)) * 1e3
for name, n, half in (("tridiagonal 8M", 8_000_000, 1), ("band of 33, 2M rows", 2_000_000, 16), ("band of 65, 1M rows", 1_000_000, 32)):
    i = np.repeat(np.arange(n, dtype=np.int64), 2 * half + 1)
    c = i + np.tile(np.arange(-half, half + 1, dtype=np.int64), n)
    ok = (c >= 0) & (c < n)
    rows, cols = i[ok], c[ok].astype(np.int32)
    lens = np.bincount(rows, minlength=n)
    rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    vals = np.random.default_rng(1).uniform(-1, 1, rows.size)
    A = ctx.csr(n, n, rp, cols, vals)
    x, y = ctx.gen_vector(n, seed=1), ctx.vector(n); y.fill(0.0)
    print(f"{name}: CSR AUTO kernel {A.info.kernel} (ELL copy variant {A.get_param('ell_copy_variant')}): {timed(A, x, y):.1f} us", end="")
    A.set_kernel(4); print(f" | panel {timed(A, x, y):.1f}", end="")
    E = ctx.csr_to_ell(A)
    print(f" | ELL handle AUTO variant {E.get_param('ell_variant')}: {timed(E, x, y):.1f}", end="")
    E.set_param("ell_dia_order", 1); print(f" | DIA order ({E.get_param('ell_non_conforming_rows')} rows aside): {timed(E, x, y):.1f}", end="")
    E.set_param("ell_dia_order", 0); E.set_kernel(1, 2); print(f" | two rows per lane: {timed(E, x, y):.1f} us")
    del A, E, x, y
