#!/usr/bin/env python3
"""bench.py — the reference's headline measurement on MI355X: fp64 CSR `y += A*x` throughput.

One "step" = one application of the hot path (CSRMatrixMatVector, reference src/mat_vec.cpp:44-67)
to the whole synthetic matrix, inputs resident in HBM (the reference's timed loop likewise runs on
arrays already in memory: main.cpp:66-69; its NUMA driver builds the x replicas before the loop,
src/mat_vec.cpp:266 vs :271).

Workload at N GPUs (weak scaling, BASELINE.json configs[1] and configs[4]): every rank owns 10M rows
x 32 entries/row of a (10M*N x 10M*N) uniform-random matrix, row-range partitioned like the reference's
NUMA driver (src/mat_vec.cpp:240-268): rebased row_ptr, global column indices, full x replica per GPU.
The x replica is assembled from the ranks' own slices by an RCCL all-gather over xGMI.

    python bench.py                                  # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(fmt: str, nrow: int, ncol: int, nnz: int, k: int = 0) -> int:
    """x counted once, y read + written (the op accumulates)."""
    if fmt == "csr":
        return 12 * nnz + 4 * (nrow + 1) + 8 * ncol + 16 * nrow
    if fmt == "ell":
        return 12 * nrow * k + 8 * ncol + 16 * nrow
    if fmt == "coo":
        return 16 * nnz + 8 * ncol + 16 * nrow
    raise ValueError(fmt)


def cpu_baseline(args, synth, nrow_total: int) -> dict:
    """Time the reference's own OpenMP CSR loop (oracle/_ref, built from the reference sources) — or, if
    that library did not travel, our restatement of it — on a bounded sample of the SAME matrix: the first
    `sample_rows` rows (regenerated bit-exactly on the host by the numpy twin of the device generator),
    with the full x.  The gather footprint per entry (all of x) is therefore the benchmark's."""
    import numpy as np

    # the host share that goes with one GPU of the box is 16 cores; more OpenMP threads than that only
    # oversubscribe (measured: 256 threads = 0.8 GFLOP/s, slower than 8 threads in the survey container)
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    cores = max(1, min(allowed, args.cpu_threads))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as ol  # the checker / baseline only — never on the product path

    m = min(args.cpu_sample_rows, args.n)
    t0 = time.perf_counter()
    row_ptr, col, val = synth.csr_uniform(0, m, nrow_total, args.k, band=args.band, seed=args.seed)
    x = synth.vec_uniform(nrow_total, seed=args.seed)
    y = np.zeros(m)
    gen_s = time.perf_counter() - t0
    kind = "reference"
    try:
        ref = ol.load_ref()
        p = ol._p

        def run():
            ref.ref_csr_spmv(m, nrow_total, p(row_ptr), p(col), p(val), p(x), p(y))
    except OSError:
        kind = "port"
        orc = ol.load_oracle()

        def run():
            ol.csr_spmv_omp(orc, row_ptr, col, val, x, y)

    run()  # warm-up + first touch
    reps, t_total = 0, 0.0
    while reps < 50 and t_total < args.cpu_seconds:
        t = time.perf_counter()
        run()
        t_total += time.perf_counter() - t
        reps += 1
    nnz = int(row_ptr[-1])
    return {
        "value": round(2.0 * nnz * reps / t_total / 1e9, 4),
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": kind,
        "sample": f"rows [0,{m}) of the benchmark matrix ({nnz} entries, full x of {nrow_total}), {reps} reps of "
                  f"CSRMatrixMatVector with OMP_NUM_THREADS={cores}; {gen_s:.1f}s to regenerate the rows on the host",
        "ms_per_apply": round(1e3 * t_total / reps, 3),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)  # NUM_TEST, main.cpp:16
    ap.add_argument("--warmup", type=int, default=5)
    # (no short spellings such as --n: torch.distributed.run would claim them as abbreviations of its own options)
    ap.add_argument("--rows", dest="n", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--per-row", dest="k", type=int, default=32, help="entries per row")
    ap.add_argument("--band", type=int, default=0, help="0 = uniform columns; >0 = random within a band of this width")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kernel", type=int, default=0, help="spmv_csr_kernel id (0 = auto)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per row for the vector kernel (0 = auto)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU baseline (host share of one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    capi, synth = pkg.capi, pkg.synth
    import importlib

    shard = importlib.import_module("arm_spmv_amd.dist")  # row-range sharding + the x all-gather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run (see docstring)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # SPMV_BENCH_BACKEND=gloo rehearses the N>1 control flow with several ranks on ONE GPU (RCCL needs one GPU per
    # rank); the measured configuration is always the default: nccl (= RCCL on ROCm), rank r on GPU r
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = capi.Context(dev_index, stream=stream.cuda_stream)
        n, k = args.n, args.k
        ncol = n * world
        row_begin, row_end = shard.shard_rows(ncol, world, rank)  # equal rows per rank (src/mat_vec.cpp:245-246)
        t_setup = time.perf_counter()
        A = ctx.gen_csr_uniform(row_begin, row_end, ncol, k, band=args.band, seed=args.seed)
        if args.kernel or args.lanes:
            A.set_kernel(args.kernel, args.lanes)
        if args.flags:
            A.set_flags(args.flags)
        info = A.info
        ctx.sync()
        setup_s = time.perf_counter() - t_setup  # generation + analysis + layout + trials: one-off, outside the timed region

        # x: every rank draws its own slice; the replica is assembled by an RCCL all-gather over xGMI
        x_full = torch.empty(ncol, dtype=torch.float64, device=dev)
        x_own = x_full[row_begin:row_end]
        vx_own = ctx.wrap_vector(x_own, n)
        capi._check(ctx._lib.spmv_gen_vec_uniform(ctx.h, vx_own.h, row_begin, args.seed))
        allgather_ms = None
        if world > 1:
            x_send = x_own.clone()
            shard.allgather_x(x_full, x_send, ncol)
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        vx, vy = ctx.wrap_vector(x_full, ncol), ctx.wrap_vector(y, n)

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            ctx.apply(A, vx, vy)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(args.steps):
            ctx.apply(A, vx, vy)
        ev1.record(stream)
        barrier()
        wall_s = time.perf_counter() - t0
        kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the stream the kernel runs on

        # secondary: the same loop with the x exchange charged to every step (solver-realistic)
        exch_s = None
        if world > 1:
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                shard.allgather_x(x_full, x_send, ncol)
                ctx.apply(A, vx, vy)
            barrier()
            exch_s = time.perf_counter() - t1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                shard.allgather_x(x_full, x_send, ncol)
            e1.record(stream)
            torch.cuda.synchronize()
            allgather_ms = e0.elapsed_time(e1) / 10

        times = torch.tensor([wall_s, kernel_ms, exch_s or 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(times, op=dist.ReduceOp.MAX)
        wall_s, kernel_ms, exch_max = (float(t) for t in times.tolist())

    if rank == 0:
        nnz_rank = int(info.nnz)
        nnz_total = nnz_rank * world
        gflops = 2.0 * nnz_total * args.steps / wall_s / 1e9
        bytes_launch = algorithmic_bytes("csr", n, ncol, nnz_rank)
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "pmc_traffic.json"
        wl_key = f"csr_n{n}_k{k}_band{args.band}_ncol{ncol}"
        if tfile.exists():
            traffic = json.loads(tfile.read_text()).get(wl_key, {}).get("hbm_bytes_per_launch")
        kernel_names = {1: "csr_vector_kernel", 2: "csr_ldswin_kernel", 3: "csr_scalar_kernel", 4: "csr_panel_kernel"}
        panel = None
        if int(info.kernel) == 4:
            panel = {k: A.get_param("panel_" + k) for k in ("rows", "width", "groups", "layout", "unroll", "pipe", "sync", "stagger", "pace_ns", "skew", "bytes")}
        out = {
            "metric": "SpMV GFLOP/s + achieved HBM GB/s (% roofline), fp64 CSR, 1/2/4/8 MI355X",
            "value": round(gflops, 3),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall_s / args.steps, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"fp64 CSR y+=A*x, {n} rows/GPU x {k} entries/row, "
                            + ("uniform-random columns" if args.band == 0 else f"columns random in a band of {args.band}")
                            + f" over {ncol} columns (BASELINE configs[{1 if world == 1 else 4}])",
                "rows_per_gpu": n,
                "nnz_per_row": k,
                "nnz_total": nnz_total,
                "ncol": ncol,
                "band": args.band,
                "seed": args.seed,
                "partition": f"row-range x{world}, full x replica per GPU (src/mat_vec.cpp:240-268)",
                "x_exchange": "static replica, all-gathered once before the timed loop (as src/mat_vec.cpp:266 vs :271)",
                "kernel": kernel_names.get(int(info.kernel), str(info.kernel)),
                "lanes_per_row": int(info.lanes_per_row),
                "setup_seconds": round(setup_s, 3),
                "panel_layout": panel,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_launch,
                "kernel_ms": round(kernel_ms, 5),
            },
        }
        if world > 1:
            out["with_x_allgather_each_step"] = {
                "value": round(2.0 * nnz_total * args.steps / exch_max / 1e9, 3),
                "unit": "GFLOP/s",
                "allgather_ms": round(allgather_ms, 4) if allgather_ms else None,
                "bytes_per_rank": 8 * n,
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, synth, ncol)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
