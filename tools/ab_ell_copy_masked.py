import sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from __graft_entry__ import load_package
capi = load_package().capi
import importlib.util
spec = importlib.util.spec_from_file_location("sw", "/root/repo/tools/sweep_structures.py"); sw = importlib.util.module_from_spec(spec); spec.loader.exec_module(sw)
ctx = capi.Context(0)
def timed(A, x, y):
    return sw.timed(ctx, A, x, y, 20)
cases = {"stencil2d 2048": lambda: sw.stencil((2048, 2048), 5), "stencil3d 160": lambda: sw.stencil((160, 160, 160), 7)}
for name, build in cases.items():
    nrow, ncol, r, c, v = build()
    ln = np.bincount(r, minlength=nrow); rp = np.concatenate(([0], np.cumsum(ln))).astype(np.int32)
    A = ctx.csr(nrow, ncol, rp, c, v)
    A.set_kernel(8)
    x, y = ctx.vector_from(np.random.default_rng(1).uniform(0, 1, ncol)), ctx.vector(nrow)
    y.fill(0.0)
    print(f"{os.environ.get('SPMV_HIP_SO', 'HEAD'):40s} {name}: ELL copy variant {A.get_param('ell_copy_variant')} {timed(A, x, y)*1e3:.2f} us")
