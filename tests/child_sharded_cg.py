"""Child process of tests/test_gpu_solver.py::test_sharded_cg_with_the_engine_as_local_ops (GPU box only)."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main():
    torch.cuda.init()  # torch's HIP runtime first, then the engine (same order as bench.py)
    dev = torch.device("cuda", 0)
    from __graft_entry__ import load_package

    import oracle_lib as ol
    from test_gpu_solver import _spd_random

    pkg = load_package()
    dmod = importlib.import_module("arm_spmv_amd.dist")
    orc = ol.load_oracle()
    ctx = pkg.capi.Context(0)
    n, rp, cc, cv = _spd_random(50_000, 6, 9)
    A = ctx.csr(n, n, rp, cc, cv)
    b_host = np.random.default_rng(3).uniform(-1, 1, n)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        b_t = torch.from_numpy(b_host).to(dev)
        x_t = torch.zeros(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        iters, relres = dmod.cg_sharded(dmod.HipShardOps(ctx, A), b_t, x_t, n, max_iter=500, rel_tol=1e-9)
        ctx.sync()
        sol = x_t.cpu().numpy()
        # the overlapped variant: at world 1 every column is the rank's own (the outside part is empty)
        ops2 = dmod.HipShardOps(ctx, A)
        ops2.enable_overlap(0, n)
        x2_t = torch.zeros(n, dtype=torch.float64, device=dev)
        iters_o, relres_o = dmod.cg_sharded(ops2, b_t, x2_t, n, max_iter=500, rel_tol=1e-9)
        ctx.sync()
        assert abs(iters_o - iters) <= 2 and relres_o <= 1e-9, (iters_o, iters, relres_o)
        assert float((x2_t - x_t).abs().max()) <= 1e-7 * float(x_t.abs().max())
        # block-Jacobi preconditioner: one symmetric Gauss-Seidel sweep on the rank's own diagonal block (at world 1: the
        # whole matrix): fewer iterations, the same solution
        ops3 = dmod.HipShardOps(ctx, A)
        ops3.enable_symgs(0, n)
        ops3.use_overlap = False
        x3_t = torch.zeros(n, dtype=torch.float64, device=dev)
        iters_p, relres_p = dmod.cg_sharded(ops3, b_t, x3_t, n, max_iter=500, rel_tol=1e-9)
        ctx.sync()
        assert relres_p <= 1e-9 and iters_p < iters, (iters_p, iters, relres_p)
        assert float((x3_t - x_t).abs().max()) <= 1e-6 * float(x_t.abs().max())
        # the same through the single-device solver with the same preconditioner
        x4 = ctx.vector(n)
        x4.fill(0.0)
        iters4, _ = ctx.cg(A, ctx.vector_from(b_host), x4, max_iter=500, rel_tol=1e-9, symgs=True)
        assert abs(iters_p - iters4) <= 2, (iters_p, iters4)
    finally:
        dist.destroy_process_group()
    ax = np.zeros(n)
    ol.csr_spmv(orc, rp, cc, cv, sol, ax)
    true_res = np.linalg.norm(b_host - ax) / np.linalg.norm(b_host)
    assert relres <= 1e-9 and true_res <= 2e-8, (relres, true_res)
    x2 = ctx.vector(n)
    x2.fill(0.0)
    iters2, _ = ctx.cg(A, ctx.vector_from(b_host), x2, max_iter=500, rel_tol=1e-9)
    assert abs(iters - iters2) <= 2, (iters, iters2)
    assert np.max(np.abs(x2.download() - sol)) <= 1e-7 * np.max(np.abs(sol))
    print(f"SHARDED_CG_OK iters={iters} single_device_iters={iters2} with_block_symgs={iters_p} true_residual={true_res:.3e}")


if __name__ == "__main__":
    main()
