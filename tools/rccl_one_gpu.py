#!/usr/bin/env python3
"""tools/rccl_one_gpu.py — every RCCL call of the engine's exchange (spmv_comm_*) that one GPU can make: SPMV_COMM=rccl forces the
RCCL transport with ONE participant (ncclCommInitAll(1), the self-check's ncclAllGather and group of ncclBroadcast, the all-gather
of a 10M-entry vector).  Run under `rocprofv3 --kernel-trace --stats -- python3 tools/rccl_one_gpu.py` to see RCCL's kernels."""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["SPMV_COMM"] = "rccl"
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
ctx = capi.Context(0)
comm = capi.Comm([ctx])
print("transport:", comm.backend)
n = 10_000_000
x = ctx.gen_vector(n, seed=3)
before = x.download()
for _ in range(5):
    comm.allgather([x], np.array([0, n], dtype=np.int64))
ctx.sync()
assert np.array_equal(x.download(), before) and comm.backend == "rccl"
print("rccl one-participant all-gather x 5: OK")
