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
