"""N > 1 path on CPU: world_size 2 and 3, gloo backend.  The collective/sharding logic is the product's
(arm-spmv_amd/dist.py + spmv_partition_rows); the per-shard product is the oracle here (the HIP engine needs a GPU),
and the assembled y must equal the unsharded oracle result bit for bit (row partitioning keeps each row's order)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nrow, k, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from __graft_entry__ import load_package
    import importlib

    pkg = load_package()
    dmod = importlib.import_module("arm_spmv_amd.dist")
    import oracle_lib as ol

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = ol.load_oracle()
        rp, col, val = pkg.synth.csr_uniform(0, nrow, nrow, k, seed=99)
        x = pkg.synth.vec_uniform(nrow, seed=99)
        b, e = dmod.shard_rows(nrow, world, rank)
        assert (b, e) == ol.partition_rows(orc, nrow, world, rank)
        # shard exactly as the NUMA driver does: rebased row_ptr, global columns
        srp = ol.csr_shard_row_ptr(orc, rp, b, e)
        scol, sval = col[rp[b]:rp[e]], val[rp[b]:rp[e]]
        # x: every rank owns its slice, the replica comes from the all-gather
        x_full = torch.full((nrow,), float("nan"), dtype=torch.float64)
        dmod.allgather_x(x_full, torch.from_numpy(x[b:e].copy()), nrow)
        assert np.array_equal(x_full.numpy(), x)
        y_own = np.zeros(e - b)
        for _ in range(3):  # accumulate like the timed loop
            ol.csr_spmv(orc, srp, np.ascontiguousarray(scol), np.ascontiguousarray(sval), x_full.numpy(), y_own)
        y_full = dmod.concatenate_y(torch.from_numpy(y_own), nrow).numpy()
        ref = np.zeros(nrow)
        for _ in range(3):
            ol.csr_spmv(orc, rp, col, val, x, ref)
        t = dmod.max_over_ranks(float(rank + 1), torch.device("cpu"))
        q.put((rank, bool(np.array_equal(y_full, ref)), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nrow", [(2, 1000), (2, 1001), (3, 1000)])
def test_row_sharded_product_over_gloo(world, nrow):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nrow, 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    assert all(r[1] for r in results), "assembled y differs from the unsharded product"
    assert all(r[2] == float(world) for r in results)


def _worker_balanced(rank, world, port, nrow, q):
    """the entry-balanced partition end to end: power-law rows SORTED BY LENGTH (every heavy row in the first ranks' range under
    equal rows), bounds from spmv_partition_rows_balanced, ragged x slices through the broadcast path, y concatenated"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from __graft_entry__ import load_package
    import importlib

    pkg = load_package()
    dmod = importlib.import_module("arm_spmv_amd.dist")
    import oracle_lib as ol

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = ol.load_oracle()
        rows, col, val = pkg.synth.coo_powerlaw(nrow, nrow, 256, seed=5, sorted_by_length=True)
        rp = np.zeros(nrow + 1, np.int64)
        np.add.at(rp, rows.astype(np.int64) + 1, 1)
        rp = np.cumsum(rp)
        x = pkg.synth.vec_uniform(nrow, seed=5)
        bounds = dmod.balanced_bounds(rp, world)
        b, e = bounds[rank]
        share = [int(rp[be[1]] - rp[be[0]]) for be in bounds]
        equal = [int(rp[be[1]] - rp[be[0]]) for be in dmod.all_bounds(nrow, world)]
        rp32 = rp.astype(np.int32)
        srp = ol.csr_shard_row_ptr(orc, rp32, b, e)
        scol, sval = np.ascontiguousarray(col[rp[b]:rp[e]]), np.ascontiguousarray(val[rp[b]:rp[e]])
        x_full = torch.full((nrow,), float("nan"), dtype=torch.float64)
        dmod.allgather_x(x_full, torch.from_numpy(x[b:e].copy()), nrow, bounds=bounds)
        assert np.array_equal(x_full.numpy(), x)
        y_own = np.zeros(e - b)
        ol.csr_spmv(orc, srp, scol, sval, x_full.numpy(), y_own)
        y_full = dmod.concatenate_y(torch.from_numpy(y_own), nrow, bounds=bounds).numpy()
        ref = np.zeros(nrow)
        ol.csr_spmv(orc, rp32, col, val, x, ref)
        bad = None
        try:
            dmod.allgather_x(x_full, torch.from_numpy(x[b:e].copy()), nrow, bounds=[(0, 1)] * world)
        except ValueError as err:
            bad = str(err)
        q.put((rank, bool(np.array_equal(y_full, ref)), max(share) / (sum(share) / world), max(equal) / (sum(equal) / world), bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_entry_balanced_partition_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_balanced, args=(r, world, port, 20_000, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in results), "assembled y differs from the unsharded product"
    for _, _, balanced, by_rows, bad in results:
        assert balanced <= 1.02 and by_rows >= 1.3, (balanced, by_rows)  # the skew is real and the balanced split removes it
        assert bad and "do not tile" in bad


class _OracleOps:
    """CPU stand-in for dist.HipShardOps in the gloo tests: the oracle's product and numpy BLAS-1"""

    def __init__(self, orc, ol, srp, scol, sval):
        self.orc, self.ol, self.m = orc, ol, (srp, scol, sval)

    def product_dot(self, p_full, w_own, q_own):
        q = q_own.numpy()
        q[:] = 0.0
        self.ol.csr_spmv(self.orc, *self.m, p_full.numpy(), q)
        return float(np.dot(w_own.numpy(), q))

    def axpby(self, alpha, x, beta, y, w):
        w.numpy()[:] = alpha * x.numpy() + beta * y.numpy()

    # overlap variant (dist.HipShardOps.enable_overlap): the shard split by column range with numpy
    def enable_overlap(self, c0, c1):
        srp, scol, sval = self.m
        rows = np.repeat(np.arange(len(srp) - 1), np.diff(srp))
        inside = (scol >= c0) & (scol < c1)

        def part(mask, rebase):
            rp = np.zeros(len(srp), np.int64)
            np.add.at(rp, rows[mask] + 1, 1)
            return (np.cumsum(rp).astype(np.int32), np.ascontiguousarray(scol[mask] - rebase, dtype=np.int32),
                    np.ascontiguousarray(sval[mask]))

        self.m_in, self.m_out = part(inside, c0), part(~inside, 0)
        self.overlap = True

    def begin_local(self, p_own, q_own):
        q = q_own.numpy()
        q[:] = 0.0
        self.ol.csr_spmv(self.orc, *self.m_in, p_own.numpy(), q)

    def finish_remote_dot(self, p_full, w_own, q_own):
        q = q_own.numpy()
        self.ol.csr_spmv(self.orc, *self.m_out, p_full.numpy(), q)
        return float(np.dot(w_own.numpy(), q))

    def dot(self, a, b):
        return float(np.dot(a.numpy(), b.numpy()))

    def sync(self):
        pass

    # block-Jacobi preconditioner (dist.HipShardOps.enable_symgs): the oracle's symmetric Gauss-Seidel sweep, in the
    # matrix's own row order, on the rank's own diagonal block
    def enable_symgs(self, c0, c1, sweeps=1):
        if not hasattr(self, "m_in"):
            self.enable_overlap(c0, c1)
        self.use_overlap = False
        self._gs_sweeps = sweeps
        self.precondition = self._precondition

    def _precondition(self, r_own, z_own):
        z = z_own.numpy()
        z[:] = 0.0
        assert self.ol.symgs(self.orc, *self.m_in, r_own.numpy(), z, self._gs_sweeps) == 0


def _cg_worker(rank, world, port, m, q, overlap=False, symgs=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from __graft_entry__ import load_package
    import importlib

    load_package()
    dmod = importlib.import_module("arm_spmv_amd.dist")
    import oracle_lib as ol

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = ol.load_oracle()
        # 1-D Laplacian plus a wrap-around coupling: symmetric positive definite, rows reach into other shards
        n = m
        rows = np.repeat(np.arange(n), 3)
        cols = np.stack([(np.arange(n) - 1) % n, np.arange(n), (np.arange(n) + 1) % n], 1).ravel()
        vals = np.tile(np.array([-1.0, 2.5, -1.0]), n)
        rp = np.arange(0, 3 * n + 1, 3, dtype=np.int32)
        col, val = cols.astype(np.int32), vals
        bvec = np.random.default_rng(4).uniform(-1, 1, n)
        b, e = dmod.shard_rows(n, world, rank)
        srp = ol.csr_shard_row_ptr(orc, rp, b, e)
        ops = _OracleOps(orc, ol, srp, np.ascontiguousarray(col[rp[b]:rp[e]]), np.ascontiguousarray(val[rp[b]:rp[e]]))
        if overlap:
            ops.enable_overlap(b, e)
        if symgs:
            ops.enable_symgs(b, e)
        x_own = torch.zeros(e - b, dtype=torch.float64)
        iters, relres = dmod.cg_sharded(ops, torch.from_numpy(bvec[b:e].copy()), x_own, n, max_iter=500, rel_tol=1e-10)
        x_full = dmod.concatenate_y(x_own, n).numpy()
        ax = np.zeros(n)
        ol.csr_spmv(orc, rp, col, val, x_full, ax)
        true_res = float(np.linalg.norm(bvec - ax) / np.linalg.norm(bvec))
        q.put((rank, iters, relres, true_res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,m,overlap,symgs", [(2, 600, False, False), (3, 601, False, False), (2, 600, True, False), (3, 601, True, False),
                                                   (2, 600, False, True), (3, 601, False, True)])
def test_sharded_cg_over_gloo(world, m, overlap, symgs):
    """dist.cg_sharded: one all-gather of the direction + two scalar all-reduces per iteration; with
    `symgs` the block-Jacobi preconditioner (a symmetric Gauss-Seidel sweep on every rank's own diagonal block, no
    exchange) and one more all-reduce"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cg_worker, args=(r, world, port, m, q, overlap, symgs)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({(r[1], r[2]) for r in results}) == 1, "ranks disagree on iterations / residual"
    assert all(0 < r[1] < 500 and r[2] <= 1e-10 and r[3] <= 1e-9 for r in results), results
    if symgs:  # (1-D Laplacian with diagonal 2.5: plain CG needs ~30 iterations to 1e-10, the sweep about half)
        assert results[0][1] <= 20, results


def _worker_plans(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "tests")]
    from __graft_entry__ import load_package
    import importlib

    load_package()
    dmod = importlib.import_module("arm_spmv_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cpu")
        # a blob the size of a three-node plan (16-byte header + 3 x 128): the bytes travel unchanged, whoever holds them first
        blob = bytes((7 * i + 3) % 251 for i in range(16 + 3 * 128))
        got0 = dmod.broadcast_plan(blob if rank == 0 else None, dev, src=0)
        same0 = dmod.plans_equal(got0, dev)
        last = world - 1
        got1 = dmod.broadcast_plan(blob[::-1] if rank == last else None, dev, src=last)
        same1 = dmod.plans_equal(got1, dev)
        # ranks that built different plans are told so
        differ = dmod.plans_equal(blob if rank == 0 else blob[:-1] + b"\x00", dev)
        empty = dmod.broadcast_plan(None, dev, src=0)  # nobody has one: an empty blob everywhere
        q.put((rank, got0 == blob, same0, got1 == blob[::-1], same1, differ, empty == b""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_plan_blob_broadcast_and_equality_over_gloo(world):
    """rank 0's plan reaches every rank byte for byte (dist.broadcast_plan: what `bench.py --gpus N` and the sharded drivers use so
    that ranks holding same-shape shards cannot diverge in their kernel), and plans_equal tells equal from different"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_plans, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in results:
        assert r[1] and r[2] and r[3] and r[4] and (not r[5]) and r[6], r
