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

# Multi-process GPU work on this platform shares device memory through dmabuf handles only: with the legacy IPC mode
# RCCL's (and torch's) cross-process buffer registration fails with `hipIpcGetMemHandle: invalid argument`.  The image
# exports HSA_ENABLE_IPC_MODE_LEGACY=0 already; bench.py sets it itself (before anything touches the GPU) so that an
# 8-rank launch does not depend on the caller's environment.  INTEGRATION.md section 4 says the same for applications.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


L2_PEAK_GBS = 34500.0  # the eight L2s together, 128-byte line operations (same guide, "L2 (per XCD)")


def required_bytes(kind: str, nrow: int, ncol: int, nnz: int, k: int = 0) -> int:
    """Bytes the kernel that actually ran HAS to move (its own storage, x once, y read + written) — what `roofline.frac`
    of an extra line is measured against, so that no fraction can exceed 1.  `kind`:
      csr / panel      12 B per entry + row_ptr          (the panel layout packs an entry into 12 bytes: CSR's own cost)
      ell_columns      12 B per slot                      (one lane per row, column indices read)
      ell_diagonals    8 B per slot                       (slots recognised as diagonals: no index stream)
      coo_segscan      16 B per entry                     (row, column, value)"""
    if kind in ("csr", "panel"):
        return 12 * nnz + 4 * (nrow + 1) + 8 * ncol + 16 * nrow
    if kind == "ell_columns":
        return 12 * nrow * k + 8 * ncol + 16 * nrow
    if kind == "ell_diagonals":
        return 8 * nrow * k + 8 * ncol + 16 * nrow
    if kind == "coo_segscan":
        return 16 * nnz + 8 * ncol + 16 * nrow
    raise ValueError(kind)


def measured_counters(key: str, kernel: str, layout: dict | None) -> dict:
    """HBM bytes and L2 line operations per launch from the committed PMC passes (profiles/pmc_traffic.json), or nulls
    when that file's entry was measured on another kernel or panel layout than the one that just ran — a stale
    constant must not pass for a measurement."""
    none = {"traffic": None, "l2_line_ops": None, "measured_on": None}
    tfile = ROOT / "profiles" / "pmc_traffic.json"
    if not tfile.exists():
        return none
    e = json.loads(tfile.read_text()).get(key)
    if not e:
        return none
    want = e.get("match", {})
    if want.get("kernel") and want["kernel"] not in kernel:
        return dict(none, measured_on=f"stale: counters are for {want['kernel']}, this run used {kernel}")
    for name, val in (want.get("panel_layout") or {}).items():
        if layout is None or int(layout.get(name, -1)) != int(val):
            return dict(none, measured_on=f"stale: counters are for panel {name}={val}, this run has {None if layout is None else layout.get(name)}")
    ops = None
    if e.get("tcp_tcc_read_req") and e.get("tcc_miss"):
        ops = int(e["tcp_tcc_read_req"]) + int(e["tcc_miss"])  # L1->L2 read requests + fills from the fabric
    return {"traffic": e.get("hbm_bytes_per_launch"), "l2_line_ops": ops, "measured_on": e.get("kernel")}


def host_topology() -> dict:
    """sockets / NUMA nodes / physical cores of the host and of the CPUs this process may run on"""
    import subprocess

    info = {}
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=20).stdout.splitlines():
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key in ("Model name", "Socket(s)", "NUMA node(s)", "Core(s) per socket", "Thread(s) per core", "CPU(s)"):
                info[key] = val
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = set()
    for c in allowed:
        try:
            base = Path(f"/sys/devices/system/cpu/cpu{c}/topology")
            cores.add((int((base / "physical_package_id").read_text()), int((base / "core_id").read_text())))
        except (OSError, ValueError):
            cores.add((0, c))
    return {
        "model": info.get("Model name"),
        "sockets": int(info["Socket(s)"]) if info.get("Socket(s)", "").isdigit() else None,
        "numa_nodes": int(info["NUMA node(s)"]) if info.get("NUMA node(s)", "").isdigit() else None,
        "cores_per_socket": int(info["Core(s) per socket"]) if info.get("Core(s) per socket", "").isdigit() else None,
        "threads_per_core": int(info["Thread(s) per core"]) if info.get("Thread(s) per core", "").isdigit() else None,
        "logical_cpus": int(info["CPU(s)"]) if info.get("CPU(s)", "").isdigit() else None,
        "allowed_cpus": len(allowed),
        "allowed_physical_cores": len(cores),
    }


def host_limits() -> dict:
    """what bounds the CPU leg besides the core count: the cgroup's CPU quota (a container may see 256 CPUs and be
    allowed 16 CPUs' worth of time), its cpuset, and the NUMA nodes the allowed CPUs sit on (no numactl on the box:
    read from sysfs)"""
    out = {}
    for name, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("cgroup_cpuset_effective", "/sys/fs/cgroup/cpuset.cpus.effective"),
                       ("cgroup_cpu_stat", "/sys/fs/cgroup/cpu.stat")):
        try:
            out[name] = Path(path).read_text().strip().replace("\n", "; ")[:300]
        except OSError:
            out[name] = None
    quota = None
    try:
        q, period = (out.get("cgroup_cpu_max") or "max 100000").split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except ValueError:
        pass
    if quota is None:  # cgroup v1
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / per
                out["cgroup_cpu_max"] = f"{q} {per} (v1 cfs_quota_us cfs_period_us)"
        except (OSError, ValueError):
            pass
    out["cgroup_cpu_quota_cpus"] = round(quota, 2) if quota else None
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = []
    nodes = {}
    for node in sorted(Path("/sys/devices/system/node").glob("node[0-9]*")):
        try:
            cpus = set()
            for part in (node / "cpulist").read_text().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
            nodes[node.name] = len(cpus & set(allowed))
        except (OSError, ValueError):
            pass
    out["allowed_cpus_per_numa_node"] = nodes or None
    return out


def cpu_baseline_child(args) -> dict:
    """The CPU leg of the report, in a process of its own (no torch: its OpenMP runtime and thread pools would be in
    the way; OMP_PROC_BIND / OMP_PLACES have to be in the environment before libgomp starts).  Times, on the GPU box's
    host cores, the reference's own code (oracle/_ref = the reference's sources compiled by oracle/Makefile) — or our
    restatement where that library did not travel — on a bounded sample of the SAME matrix: the first `sample_rows`
    rows (regenerated bit-exactly by the numpy twin of the device generator) with the full x, so the gather footprint
    per entry is the benchmark's.  Modes:
      openmp           CSRMatrixMatVector (src/mat_vec.cpp:44-67), one thread per allowed physical core, spread binding
      numa_reference   CSRMatrixMatVectorNuma (src/mat_vec.cpp:230-297) as it is: 50 repetitions inside, pthreads
                       re-created in every repetition (:274-281); its own "### CSR NUMA GFLOPS" line is what is reported
      numa_persistent  the same sharding with persistent pinned workers (oracle/spmv_oracle.c: orc_csr_spmv_sharded)
      single_thread_c1 BASELINE configs[0]: 10k x 10k, 16 per row, one thread, through a Matrix Market file
    """
    import ctypes as C
    import tempfile

    import numpy as np

    from __graft_entry__ import load_package

    synth = load_package().synth
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as ol  # the checker / baseline only — never on the product path

    topo = host_topology()
    facts = host_limits()  # (before libgomp starts: OMP_PROC_BIND pins the initial thread, and with it sched_getaffinity)
    threads = max(1, min(topo["allowed_physical_cores"], args.cpu_threads if args.cpu_threads > 0 else 1 << 30))
    nrow_total = args.n * max(args.gpus, 1)
    # "the same synthetic matrix as each GPU config".  The OpenMP mode - the value reported - runs on the
    # WHOLE matrix of the headline (--cpu-sample-rows 0, the default: all args.n rows, 3.84 GB for C2), regenerated bit-exactly
    # by the numpy twin of the device generator in slabs of 1M rows; the two NUMA-driver modes, which repeat 50 products
    # inside one call at ~1 GFLOP/s, run on the first --cpu-numa-rows rows of it (said so in their entries).
    m = args.n if args.cpu_sample_rows <= 0 else min(args.cpu_sample_rows, args.n)
    t0 = time.perf_counter()
    row_ptr = (np.arange(m + 1, dtype=np.int64) * args.k).astype(np.int32)
    col, val = np.empty(m * args.k, np.int32), np.empty(m * args.k, np.float64)
    slab = 1_000_000
    for r0 in range(0, m, slab):
        r1 = min(m, r0 + slab)
        _, c_, v_ = synth.csr_uniform(r0, r1, nrow_total, args.k, band=args.band, seed=args.seed)
        col[r0 * args.k:r1 * args.k], val[r0 * args.k:r1 * args.k] = c_, v_
    del c_, v_
    x = synth.vec_uniform(nrow_total, seed=args.seed)
    gen_s = time.perf_counter() - t0
    nnz = int(row_ptr[-1])
    m_numa = min(args.cpu_numa_rows, m)  # rows of the same matrix the NUMA-driver modes run on
    nnz_numa = int(row_ptr[m_numa])
    p = ol._p
    orc = ol.load_oracle()
    try:
        ref = ol.load_ref()
    except OSError:
        ref = None
    gomp = C.CDLL("libgomp.so.1")
    budget = args.cpu_seconds

    def timed(run, max_reps=50):
        run()  # warm-up + first touch
        reps, total = 0, 0.0
        while reps < max_reps and total < budget:
            t = time.perf_counter()
            run()
            total += time.perf_counter() - t
            reps += 1
        return reps, total

    modes = []
    # ---- OpenMP mode: CSRMatrixMatVector with the arrays first-touched inside the OpenMP team, a sweep over team sizes
    quota = facts.get("cgroup_cpu_quota_cpus")
    cap = topo["allowed_physical_cores"]
    if args.cpu_threads > 0:
        sweep = [threads]
    else:
        sweep = sorted({t for t in (8, 16, 32, 64, 128, cap, int(quota) if quota else 0) if 0 < t <= cap})
    orc.orc_csr_first_touch_copy.restype = None
    per_point = max(1.0, budget / max(len(sweep), 1))
    sweep_out = []
    for t_n in sweep:
        gomp.omp_set_num_threads(t_n)
        # fresh, never-written destinations: their pages are placed by the team's first touch (same static row schedule)
        d_rp, d_col, d_val = np.empty(m + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
        d_x, y = np.empty(nrow_total, np.float64), np.empty(m, np.float64)
        orc.orc_csr_first_touch_copy(C.c_int32(m), C.c_int64(nrow_total), p(row_ptr), p(col), p(val), p(x), p(d_rp), p(d_col), p(d_val), p(d_x), p(y))
        if ref is not None:
            run = lambda: ref.ref_csr_spmv(m, nrow_total, p(d_rp), p(d_col), p(d_val), p(d_x), p(y))
        else:
            run = lambda: ol.csr_spmv_omp(orc, d_rp, d_col, d_val, d_x, y)
        run()  # warm-up
        reps, total = 0, 0.0
        while reps < 50 and total < per_point:
            t = time.perf_counter()
            run()
            total += time.perf_counter() - t
            reps += 1
        sweep_out.append({"threads": t_n, "value": round(2.0 * nnz * reps / total / 1e9, 4), "ms_per_apply": round(1e3 * total / reps, 3), "reps": reps})
        del d_rp, d_col, d_val, d_x, y
    best = max(sweep_out, key=lambda e: e["value"])
    threads = best["threads"]
    openmp = {"mode": "openmp", "kind": "reference" if ref is not None else "port", "threads": threads,
              "bind": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
              "first_touch": "matrix, x and y copied into fresh arrays inside the OpenMP team (static row schedule of the product) before timing",
              "value": best["value"], "unit": "GFLOP/s", "ms_per_apply": best["ms_per_apply"], "reps": best["reps"],
              "thread_sweep": sweep_out}
    modes.append(openmp)
    budget = min(budget, 10.0)
    numa_sample = (f"rows [0,{m_numa}) of the benchmark matrix ({nnz_numa} entries, full x): this mode repeats 50 products inside one call"
                   if m_numa < m else "the whole benchmark matrix")
    # ---- the reference's NUMA driver as it is (prints its own line; 50 repetitions inside)
    if ref is not None:
        y = np.zeros(m_numa)
        sys.stdout.flush()
        with tempfile.TemporaryFile(mode="w+b") as cap:
            saved = os.dup(1)
            os.dup2(cap.fileno(), 1)
            try:
                t = time.perf_counter()
                ref.ref_csr_spmv_numa(m_numa, nrow_total, p(row_ptr), p(col), p(val), p(x), p(y), threads)
                C.CDLL(None).fflush(None)
                wall = time.perf_counter() - t
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            cap.seek(0)
            text = cap.read().decode(errors="replace")
        gf = None
        for line in text.splitlines():
            if "CSR NUMA GFLOPS" in line:
                gf = float(line.split("=")[1])
        modes.append({"mode": "numa_reference", "kind": "reference", "shards": threads, "value": gf, "unit": "GFLOP/s", "sample": numa_sample,
                      "note": "the reference's own print-out over its 50 repetitions; it re-creates its pthreads in every one "
                              "(src/mat_vec.cpp:274-281) and builds the shards inside the call", "call_seconds": round(wall, 2)})
    # ---- the same sharding with persistent pinned workers
    y = np.zeros(m_numa)
    orc.orc_csr_spmv_sharded.restype = C.c_double
    reps_p = max(3, min(50, int(budget / max(openmp["ms_per_apply"] * 1e-3 * m_numa / m, 1e-4) / 10)))
    ms = orc.orc_csr_spmv_sharded(C.c_int32(m_numa), C.c_int32(nrow_total), p(row_ptr), p(col), p(val), p(x), p(y), C.c_int32(threads), C.c_int32(reps_p))
    modes.append({"mode": "numa_persistent", "kind": "port", "shards": threads, "value": round(2.0 * nnz_numa / (ms * 1e-3) / 1e9, 4) if ms > 0 else None,
                  "unit": "GFLOP/s", "ms_per_apply": round(ms, 3), "reps": reps_p, "sample": numa_sample,
                  "note": "one pinned persistent worker per shard, private shard + x replica first-touched by its worker"})
    # ---- BASELINE configs[0]: C1 through a Matrix Market file, one thread
    n1, k1 = 10_000, 16
    rp1, c1, v1 = synth.csr_uniform(0, n1, n1, k1, seed=args.seed)
    x1 = synth.vec_uniform(n1, seed=args.seed)
    gomp.omp_set_num_threads(1)
    y1 = np.zeros(n1)
    via = "arrays (no reference library here)"
    if ref is not None:
        # the reference's own reader (src/data_io.cpp:45-105) and converting constructor (src/matrix.cpp:115-154)
        class Coo(C.Structure):  # include/matrix.h:9-16
            _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_int), ("row_ind", C.POINTER(C.c_int)),
                        ("col_ind", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_double))]

        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, "c1.mtx")
            rows1 = np.repeat(np.arange(n1), np.diff(rp1))
            with open(path, "w") as f:
                f.write("%%MatrixMarket matrix coordinate real general\n")
                f.write(f"{n1} {n1} {len(v1)}\n")
                np.savetxt(f, np.column_stack([rows1 + 1, c1 + 1, v1]), fmt=["%d", "%d", "%.17g"])
            A = Coo()
            sys.stdout.flush()
            saved = os.dup(1)
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 1)
            try:
                ref._Z13COOMatrixReadPKcR9COOMatrix(path.encode(), C.byref(A))
                C.CDLL(None).fflush(None)
            finally:
                os.dup2(saved, 1)
                os.close(saved)
                os.close(devnull)
        rp1 = np.zeros(n1 + 1, np.int32)
        c1 = np.zeros(A.nnz, np.int32)
        v1 = np.zeros(A.nnz, np.float64)
        ref.ref_coo_to_csr(A.nrow, A.ncol, A.nnz, A.row_ind, A.col_ind, A.values, p(rp1), p(c1), p(v1))
        via = "read back from a Matrix Market file by the reference's COOMatrixRead, CSRMatrix(COO) by the reference"
    run1 = (lambda: ref.ref_csr_spmv(n1, n1, p(rp1), p(c1), p(v1), p(x1), p(y1))) if ref is not None else (lambda: ol.csr_spmv(orc, rp1, c1, v1, x1, y1))
    reps1, total1 = timed(run1, max_reps=2000)
    modes.append({"mode": "single_thread_c1", "kind": "reference" if ref is not None else "port", "threads": 1,
                  "value": round(2.0 * int(rp1[-1]) * reps1 / total1 / 1e9, 4), "unit": "GFLOP/s", "ms_per_apply": round(1e3 * total1 / reps1, 5),
                  "workload": f"{n1} x {n1}, {k1} per row (BASELINE configs[0]); {via}"})
    return {
        "value": openmp["value"],
        "unit": "GFLOP/s",
        "cores": threads,
        "kind": openmp["kind"],
        "sample": (f"the WHOLE benchmark matrix ({m} rows, {nnz} entries, x of {nrow_total}: BASELINE.md section 4's 'same synthetic matrix')" if m == args.n
                   else f"rows [0,{m}) of the benchmark matrix ({nnz} entries, full x of {nrow_total})") + "; headline value = the OpenMP mode, best of a "
                  f"sweep over team sizes {[e['threads'] for e in sweep_out]} ({openmp['reps']} reps of CSRMatrixMatVector with {threads} threads; "
                  f"{topo['allowed_physical_cores']} physical cores allowed, cgroup CPU quota {facts.get('cgroup_cpu_max')}); arrays first-touched inside "
                  f"the OpenMP team; {gen_s:.1f}s to regenerate the rows on the host",
        "ms_per_apply": openmp["ms_per_apply"],
        "host": topo,
        "host_limits": facts,
        "compiler": "g++ -O2 -fopenmp -DUSE_OPENMP (the reference's flags, oracle/Makefile); restatement: gcc -O2 -fopenmp -ffp-contract=off",
        "modes": modes,
    }


def cpu_baseline(args) -> dict:
    """runs cpu_baseline_child() in a fresh interpreter (see there) and returns its JSON"""
    import subprocess

    env = dict(os.environ)
    env.update({"OMP_PROC_BIND": "spread", "OMP_PLACES": "cores", "OMP_DYNAMIC": "false"})
    env.pop("OMP_NUM_THREADS", None)
    cmd = [sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-child", "--rows", str(args.n), "--per-row", str(args.k),
           "--band", str(args.band), "--seed", str(args.seed), "--cpu-sample-rows", str(args.cpu_sample_rows), "--cpu-numa-rows", str(args.cpu_numa_rows),
           "--cpu-seconds", str(args.cpu_seconds), "--cpu-threads", str(args.cpu_threads), "--gpus", str(args.gpus)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "reference", "sample": "the CPU leg failed: " + (r.stderr or r.stdout)[-400:]}
    except subprocess.SubprocessError as e:
        return {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "reference", "sample": f"the CPU leg failed: {e}"}


def pmc_child(args) -> None:
    """What the counter passes profile (`rocprofv3 --pmc ... -- python3 bench.py --pmc-child`): the headline matrix built the
    same way, one warm-up and three products through the engine, nothing else (no torch: the engine allocates its own
    vectors).  Prints the kernel and panel layout that ran, so that the parent can tell a child whose trial picked another
    instance from one that measured its own."""
    from __graft_entry__ import load_package

    capi = load_package().capi
    ctx = capi.Context(0)
    A = ctx.gen_csr_uniform(0, args.n, args.n, args.k, band=args.band, seed=args.seed)
    if args.kernel or args.lanes:
        A.set_kernel(args.kernel, args.lanes)
    if args.flags:
        A.set_flags(args.flags)
    x, y = ctx.gen_vector(args.n, seed=args.seed), ctx.vector(args.n)
    y.fill(0.0)
    for _ in range(4):
        ctx.apply(A, x, y)
    ctx.sync()
    layout = None
    if int(A.info.kernel) == 4:
        layout = {name: A.get_param("panel_" + name) for name in ("layout", "unroll", "pipe", "sync")}
    print(json.dumps({"pmc_child": True, "kernel_id": int(A.info.kernel), "panel_layout": layout}), flush=True)


PRODUCT_KERNELS = {4: ("csr_panel_pp_kernel",), 5: ("tp_expand_kernel", "tp_reduce_kernel"), 1: ("csr_vector_kernel",), 2: ("csr_ldswin_kernel",),
                   3: ("csr_scalar_kernel",)}


def pmc_mean_of_products(rows: list, kernel_id: int):
    """From the rows of one counter of a rocprofv3 counter_collection.csv (dicts with Kernel_Name, Dispatch_Id, Counter_Value):
    per kernel of the product `kernel_id` runs, the mean counter value of its LAST THREE dispatches (the child's three products
    after the warm-up; the panel kernel's trial launches - `true` as the fourth template argument - are not products), and the
    short name of the last kernel.  Returns ({kernel: mean}, name) or a string saying what is missing."""
    per_kernel, shown = {}, None
    for name in PRODUCT_KERNELS.get(kernel_id, ()):
        mine = [row for row in rows if name in row["Kernel_Name"]]
        if name == "csr_panel_pp_kernel":
            # csr_panel_pp_kernel<U, LAYOUT, ORDER, TRIAL, SYNC> (rounds 2-4: csr_panel_kernel<U, LAYOUT, PIPE, TRIAL, TRACE, SYNC>):
            # the fourth argument says "a build-time trial launch" in both
            mine = [row for row in rows if "csr_panel_kernel<" in row["Kernel_Name"] or "csr_panel_pp_kernel<" in row["Kernel_Name"]]
            mine = [row for row in mine if row["Kernel_Name"].split("_kernel<")[1].split(">")[0].split(",")[3].strip() == "false"]
        mine.sort(key=lambda row: int(row["Dispatch_Id"]))
        last = mine[-3:]
        if len(last) < 3:
            return f"fewer than 3 dispatches of {name}"
        if any(row["Kernel_Name"] != last[-1]["Kernel_Name"] for row in last):
            return f"the last three product dispatches of {name} are not one kernel"
        per_kernel[name] = sum(float(row["Counter_Value"]) for row in last) / len(last)
        shown = last[-1]["Kernel_Name"].split("(anonymous namespace)::", 1)[-1].split("(")[0]
    if not per_kernel:
        return f"no product kernel known for kernel id {kernel_id}"
    return per_kernel, shown


def live_counters(args) -> dict:
    """HBM-side traffic of the dominant kernel measured NOW, by this very bench.py: two rocprofv3 --pmc passes (FETCH_SIZE
    and WRITE_SIZE separately: they do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') over `--pmc-child`,
    run before this process touches the GPU.  Per launch: 2 x FETCH_SIZE (gfx950 tallies its 128-byte fabric reads at
    64 B: the guide's correction) + WRITE_SIZE, both KB x 1024, from the product launches only (trial launches of the
    panel kernel carry `true` as their fourth template argument).  Returns {"traffic": bytes or None, ...}; on any failure
    the caller falls back to the stamped constants of profiles/pmc_traffic.json."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    out = {"traffic": None, "source": None, "kernel": None, "child": None, "error": None}
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return dict(out, error="rocprofv3 not found")
    if any(k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return dict(out, error="bench.py itself runs under a profiler: no nested counter passes")
    child = [sys.executable, str(Path(__file__).resolve()), "--pmc-child", "--rows", str(args.n), "--per-row", str(args.k), "--band", str(args.band),
             "--seed", str(args.seed), "--kernel", str(args.kernel), "--lanes", str(args.lanes), "--flags", str(args.flags)]
    sums = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            try:
                r = subprocess.run([rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=240)
            except (OSError, subprocess.SubprocessError) as e:
                return dict(out, error=f"{counter} pass: {e}")
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and "pmc_child" in ln]
            if r.returncode != 0 or not line:
                return dict(out, error=f"{counter} pass failed (rc {r.returncode}): " + (r.stderr or r.stdout)[-300:])
            out["child"] = json.loads(line[-1])
            rows = []
            for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
                rows += [row for row in csv.DictReader(open(f)) if row["Counter_Name"] == counter]
            got = pmc_mean_of_products(rows, out["child"]["kernel_id"])
            if isinstance(got, str):
                return dict(out, error=f"{counter} pass: {got}")
            per_kernel, out["kernel"] = got
            sums[counter] = sum(per_kernel.values())
    out["traffic"] = int(round((2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024))
    out["fetch_size_kb"], out["write_size_kb"] = round(sums["FETCH_SIZE"], 1), round(sums["WRITE_SIZE"], 1)
    out["source"] = "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --pmc-child`, mean of 3 product launches"
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


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
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the headline matrix the CPU baseline's OpenMP mode runs on; 0 = all of them (BASELINE.md section 4: the same matrix)")
    ap.add_argument("--cpu-numa-rows", type=int, default=1_000_000,
                    help="rows of it the two NUMA-driver modes run on (they repeat 50 products inside one call at ~1 GFLOP/s)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = every physical core this process may use)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--keep-csr", action="store_true", help="keep col_ind / values of the CSR copy next to the panel layout")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C4 / band lines after the headline loop")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the two rocprofv3 --pmc passes; use profiles/pmc_traffic.json")
    ap.add_argument("--placement-budget-mb", type=int, default=0,
                    help="two-phase shards (x several times longer than the shard has rows: N >= 4): device memory the piece search of the product "
                         "stream may hold while it runs; 0 = 65536 at N > 1 (one slow rank sets the step of the whole job, the job has the devices "
                         "to itself, and the device's memory comes in one-class chunks of up to ~60 GB: DESIGN 4.7; since round 5 the pool is "
                         "sampled, ~1-2 s of set-up per rank instead of 4.8) and the engine's own default, 8192, at N = 1 (include/spmv_abi.h, "
                         "'twophase_placement_budget_mb')")
    ap.add_argument("--partition", choices=("rows", "nnz", "both"), default="both",
                    help="the skewed extra (C4's row lengths sorted by length, the heavy rows at one end): cut into shards by equal rows "
                         "(the reference's split, src/mat_vec.cpp:245-246), by stored entries (spmv_partition_rows_balanced), or both")
    ap.add_argument("--no-shared-plan", action="store_true", help="N > 1: every rank selects its kernel by itself (A/B against rank 0's plan broadcast)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_child:
        print(json.dumps(cpu_baseline_child(args)), flush=True)
        return
    if args.pmc_child:
        pmc_child(args)
        return
    # N = 1: the counter passes run first, while this process has not touched the GPU (one GPU process at a time)
    live = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_live_counters:
        live = live_counters(args)

    import torch
    import torch.distributed as dist

    from __graft_entry__ import load_package

    pkg = load_package()
    capi, synth = pkg.capi, pkg.synth
    import importlib

    shard = importlib.import_module("arm_spmv_amd.dist")  # row-range sharding + the x all-gather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.placement_budget_mb <= 0:
        args.placement_budget_mb = 65536 if world > 1 else 8192
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run (see docstring)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # SPMV_BENCH_BACKEND=gloo rehearses the N>1 control flow with several ranks on ONE GPU (RCCL needs one GPU per
    # rank); the measured configuration is always the default: nccl (= RCCL on ROCm), rank r on GPU r
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        args.placement_budget_mb = min(args.placement_budget_mb, 8192)  # rehearsal: the ranks share ONE device's memory
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # Under torch.distributed.run the process group exists at EVERY world size, 1 included: `torchrun --nproc-per-node 1
    # bench.py --gpus 1` runs the same RCCL calls (all-gather of x, barrier, all-reduce of the times) as the 8-GPU job,
    # which is how a one-GPU box exercises them.  A plain `python bench.py` (no RANK in the environment) has no group.
    grouped = world > 1 or "RANK" in os.environ
    if grouped:
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
        # N > 1: the ranks hold same-shape shards, and AUTO is a measurement - every rank would draw its own kernel.  Rank 0
        # builds its shard first and its PLAN (spmv_mat_get_plan: kernel, layout, tuned parameters) is broadcast; the other
        # ranks build theirs under it (spmv_ctx_set_plan: no timing launch), as the reference builds all its shards the same
        # way (src/mat_vec.cpp:240-268).  `plans_equal` on the line says what the ranks ended up with.
        shared_plan = None
        if grouped and rank != 0 and not (args.kernel or args.lanes) and not args.no_shared_plan:
            shared_plan = shard.broadcast_plan(None, dev, src=0)
            if shared_plan:
                ctx.set_plan(shared_plan)
        A = ctx.gen_csr_uniform(row_begin, row_end, ncol, k, band=args.band, seed=args.seed)
        if grouped and rank == 0 and not (args.kernel or args.lanes) and not args.no_shared_plan:
            shared_plan = shard.broadcast_plan(A.get_plan(), dev, src=0)
        ctx.set_plan(None)
        if args.kernel or args.lanes:
            A.set_kernel(args.kernel, args.lanes)
        if args.flags:
            A.set_flags(args.flags)
        info = A.info
        built_under_plan = bool(shared_plan) and rank != 0
        if int(info.kernel) == 5 and (args.placement_budget_mb != 8192 or built_under_plan):
            # The engine chose its product stream's pieces within its default budget (8 GB beyond the stream) when it built the
            # layout.  This job has the device to itself, so it grants the search more (transient: freed before the call returns);
            # one slow rank sets the step of the whole job.  Reported under config.twophase_layout.  A shard built under rank 0's
            # plan was built WITHOUT any search (where the stream lies in this device's memory is no part of a plan): it runs now.
            if args.placement_budget_mb != 8192:
                A.set_param("twophase_placement_budget_mb", args.placement_budget_mb)
            A.set_param("twophase_choose_pieces", 1)
        ctx.sync()
        setup_s = time.perf_counter() - t_setup  # generation + analysis + layout + trials: one-off, outside the timed region
        # the panel layout holds every entry once more, re-ordered: the product needs nothing else of the CSR copy but
        # row_ptr, so the handle gives col_ind / values back (memory ~1x the matrix instead of 2x)
        bytes_with_csr = A.get_param("device_bytes")
        if int(info.kernel) in (4, 5) and not args.keep_csr:
            A.set_param("panel_keep_csr", 0)
        bytes_held = A.get_param("device_bytes")
        setup_no_trial_s = None
        setup_with_plan_s = None
        my_plan = A.get_plan()
        plans_equal = shard.plans_equal(my_plan, dev) if grouped else None
        if world == 1 and int(info.kernel) == 4 and not args.no_extra:
            os.environ["SPMV_PANEL_TRIAL"] = "0"
            t1 = time.perf_counter()
            B = ctx.gen_csr_uniform(row_begin, row_end, ncol, k, band=args.band, seed=args.seed)
            ctx.sync()
            setup_no_trial_s = time.perf_counter() - t1
            del B
            os.environ.pop("SPMV_PANEL_TRIAL")
            # ... and under the plan of the handle that is measured: the same kernel, layout and tuned parameters, no timing launch
            ctx.set_plan(my_plan)
            t1 = time.perf_counter()
            B = ctx.gen_csr_uniform(row_begin, row_end, ncol, k, band=args.band, seed=args.seed)
            ctx.sync()
            setup_with_plan_s = time.perf_counter() - t1
            ctx.set_plan(None)
            assert B.get_plan() == my_plan and B.get_param("select_candidates") == 0
            del B

        # x: every rank draws its own slice; the replica is assembled by an RCCL all-gather over xGMI
        x_full = torch.empty(ncol, dtype=torch.float64, device=dev)
        x_own = x_full[row_begin:row_end]
        vx_own = ctx.wrap_vector(x_own, n)
        capi._check(ctx._lib.spmv_gen_vec_uniform(ctx.h, vx_own.h, row_begin, args.seed))
        allgather_ms = None
        if grouped:
            x_send = x_own.clone()
            shard.allgather_x(x_full, x_send, ncol)
        y = torch.zeros(n, dtype=torch.float64, device=dev)
        vx, vy = ctx.wrap_vector(x_full, ncol), ctx.wrap_vector(y, n)

        def barrier():
            torch.cuda.synchronize()
            if grouped:
                dist.barrier()
            torch.cuda.synchronize()

        def max_over_ranks(*vals):
            t = torch.tensor(list(vals), dtype=torch.float64, device=dev)
            if grouped:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t.tolist()]

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
        if grouped:
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

        # the optional last step of the sharded product: every rank's y slice gathered into the full y on every rank
        # (the reference does it for DIA only, src/mat_vec.cpp:474-477); timed by itself, never part of `value`
        y_concat_ms = None
        if grouped:
            y_full = shard.concatenate_y(y, ncol)  # warm-up (allocates the result)
            barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                y_full = shard.concatenate_y(y, ncol)
            barrier()
            (y_concat_ms,) = max_over_ranks(1e3 * (time.perf_counter() - t1) / 10)
            del y_full

        panel_names = ("rows", "width", "groups", "layout", "unroll", "pipe", "sync", "bytes")

        def panel_of(M):
            try:
                return {name: M.get_param("panel_" + name) for name in panel_names}
            except capi.SpmvError:
                return None

        def describe(fmt, M, inf, flags):
            """(kernel that runs, which bytes it has to move) of a handle"""
            kid = int(inf.kernel)
            inner_names = {1: "csr_vector_kernel", 2: "csr_ldswin_kernel", 3: "csr_scalar_kernel", 4: "csr_panel_pp_kernel", 5: "tp_expand_kernel + tp_reduce_kernel (two-phase)",
                           6: "coo_segscan_kernel (over the row-grouped entries)", 7: "long rows split off (kernels_csr_split.hip), the others through a copy", 8: "the ELL kernels over an ELL copy"}
            inner = inner_names.get(M.get_param("rowgrouped_kernel"), "csr_panel_pp_kernel") if fmt in ("ell", "coo") and kid == 4 else None
            if fmt == "ell":
                if kid == 4:
                    return f"{inner} on the row-grouped copy of the ELL slots", "panel"
                if M.get_param("ell_dia_order") and not (flags & 8):
                    return ("dia_kernel over the DIA-order (row-major) copy of the values: no column index read, x through an LDS window, "
                            f"{M.get_param('ell_non_conforming_rows')} non-conforming rows by a side kernel over the column-major arrays; +8 bytes per slot of device memory"), "ell_diagonals"
                if M.get_param("ell_diagonal_slots") and not (flags & 8):  # 8 = SPMV_FLAG_ELL_READ_COLUMNS
                    return ("ell_diag_kernel_x2 (slots recognised as diagonals: conforming rows read no column index"
                            + ("; values read from the copy in tiles of 512 rows)" if M.get_param("ell_tiled_values") else ")")), "ell_diagonals"
                return "ell_kernel_x2 (one lane per two rows, column-major slots, every column index read)", "ell_columns"
            if fmt == "coo":
                if kid == 4:
                    return f"{inner} on the row-grouped copy" + (" (12-byte packed entries)" if inner == "csr_panel_pp_kernel" else ""), "panel"
                if M.get_param("coo_column_bins"):
                    return (f"coo_segscan_bins_kernel (wavefront segmented scan over a copy of the entries in {M.get_param('coo_column_bins')} column "
                            "bins, one per XCD: each XCD gathers x from a slice that stays in its L2)"), "coo_segscan"
                return "coo_segscan_kernel (wavefront segmented scan over the entries in file order)", "coo_segscan"
            names = {1: "csr_vector_kernel", 2: "csr_ldswin_kernel", 3: "csr_scalar_kernel", 4: "csr_panel_pp_kernel",
                     5: "tp_expand_kernel + tp_reduce_kernel (two-phase)", 6: "coo_segscan_kernel (over the CSR entries and a row index per entry)",
                     7: "long rows split off (kernels_csr_split.hip), the others through a copy with a kernel of its own",
                     8: "ell_diag_kernel_x2 / ell_kernel_x2 over an ELL copy of the CSR handle (rows of nearly equal length)"}
            return names.get(kid, str(kid)), "csr"

        # After the headline loop: the other single-GPU configurations of BASELINE.json (C3: ELL, C4: COO), each once with
        # the kernel the engine picks and once with the kernel the config NAMES (column-reading ELL, COO segmented scan),
        # the band-random variant of C2 (at every world size: each rank generates its band shard) and,
        # at N = 1, the shard shape of C5.  5 warm-up + 50 applications between HIP events on the engine's stream.
        # Reported in "extra"; they never enter "value".
        extra = []
        if not args.no_extra:
            def one(name, fmt, make, tkey=None, x_vec=None, sharded=False, flags=0):
                t = time.perf_counter()
                M = make()
                if flags:
                    M.set_flags(flags)
                inf = M.info
                ctx.sync()
                t_set = time.perf_counter() - t
                vx2 = x_vec if x_vec is not None else ctx.gen_vector(int(inf.ncol), seed=args.seed)
                vy2 = ctx.vector(int(inf.nrow))
                vy2.fill(0.0)
                for _ in range(5):
                    ctx.apply(M, vx2, vy2)
                barrier()
                ms = ctx.apply_timed(M, vx2, vy2, 50)
                (ms,) = max_over_ranks(ms)
                nnz2 = int(inf.nnz)
                kk = int(inf.ell_k) if fmt == "ell" else 0
                kernel, kind = describe(fmt, M, inf, flags)
                b_fmt = algorithmic_bytes(fmt, int(inf.nrow), int(inf.ncol), nnz2, kk)
                b_req = required_bytes(kind, int(inf.nrow), int(inf.ncol), nnz2, kk)
                layout = panel_of(M) if int(inf.kernel) == 4 else None
                counters = measured_counters(tkey, kernel, layout) if tkey else {"traffic": None, "l2_line_ops": None, "measured_on": None}
                parts = world if sharded else 1
                extra.append({
                    "name": name, "format": fmt, "kernel": kernel, "nrow": int(inf.nrow), "ncol": int(inf.ncol), "nnz": nnz2 * parts,
                    "n_gpus": parts, "max_row_nnz": int(inf.max_row_nnz), "kernel_id": int(inf.kernel), "ms": round(ms, 5),
                    "value": round(2.0 * nnz2 * parts / ms / 1e6, 2), "unit": "GFLOP/s", "setup_seconds": round(t_set, 3),
                    **({"twophase_layout": {"product_stream_pieces": M.get_param("twophase_pieces"),
                                            "placement_budget_mb": M.get_param("twophase_placement_budget_mb"),
                                            "configurations_timed": M.get_param("twophase_placements_timed"),
                                            "pieces_exchanged": M.get_param("twophase_pieces_exchanged"),
                                            "as_built_over_kept": M.get_param("twophase_placement_spread") / 1000.0}} if int(inf.kernel) == 5 else {}),
                    "roofline": {"bound": "hbm", "achieved": round(b_req / ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(b_req / ms / 1e6 / HBM_PEAK_GBS, 4), "bytes_required": b_req, "bytes_required_kind": kind,
                                 "frac_of_format_bytes": round(b_fmt / ms / 1e6 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": b_fmt,
                                 "traffic": counters["traffic"],
                                 "traffic_gbs": round(counters["traffic"] / ms / 1e6, 1) if counters["traffic"] else None,
                                 "traffic_measured_on": counters["measured_on"],
                                 "per": "rank" if sharded else "launch"},
                })
                del M, vy2

            if world == 1:
                def c3():
                    return ctx.gen_ell_banded(4_000_000, 4_000_000, 64, seed=args.seed)
                one("C3: ELL N=4M, 64 per row, circulant band (BASELINE configs[2]) - the kernel the engine picks", "ell", c3,
                    tkey="ell_n4000000_k64")
                one("C3 with the kernel configs[2] names: coalesced column-major ELL, every column index read (SPMV_FLAG_ELL_READ_COLUMNS)",
                    "ell", c3, tkey="ell_n4000000_k64_columns", flags=8)

                def c4(kernel, in_place=False):
                    M = ctx.gen_coo_powerlaw(2_000_000, 2_000_000, 4096, seed=args.seed)
                    if kernel:
                        M.set_kernel(kernel, 0)
                    if in_place:
                        M.set_param("coo_column_bins", 0)
                    return M
                one("C4: COO N=2M, power-law rows up to 4096, row-sorted (BASELINE configs[3]; realised nnz reported) - the kernel the engine picks",
                    "coo", lambda: c4(0), tkey="coo_n2000000_nnz115008628")
                one("C4 with the kernel configs[3] names: COO segmented scan (spmv_mat_set_kernel VECTOR), over the entries in column bins per XCD",
                    "coo", lambda: c4(1), tkey="coo_n2000000_nnz115008628_segscan_bins")
                one("C4, the segmented scan over the entries as stored (coo_column_bins = 0): every gather of x misses the XCD's L2",
                    "coo", lambda: c4(1, True), tkey="coo_n2000000_nnz115008628_segscan")
            if args.band == 0:
                def band_shard():
                    M = ctx.gen_csr_uniform(row_begin, row_end, ncol, k, band=65536, seed=args.seed)
                    if int(M.info.kernel) in (4, 5):
                        M.set_param("panel_keep_csr", 0)
                    return M
                one(f"C2 shape, columns random in a band of 65536 around the diagonal (CSR, {n} rows per GPU x {k}, {ncol} columns"
                    + (f"; rank r holds rows r*{n}..: the band keeps its locality under sharding)" if world > 1 else ")"),
                    "csr", band_shard, tkey=f"csr_n{n}_k{k}_band65536_ncol{ncol}", x_vec=vx, sharded=world > 1)
            if world == 1 and args.band == 0:
                def c2_rows():
                    M = ctx.gen_csr_uniform(0, n, ncol, k, band=0, seed=args.seed)
                    M.set_kernel(1, 0)  # SPMV_CSR_VECTOR, lanes per row chosen from the mean row length
                    return M
                one("C2 with the kernel configs[1] names: row-parallel CSR, 2^k lanes of a wavefront per row (spmv_mat_set_kernel VECTOR) - "
                    "every gather of x misses L2 and pulls a 128-byte line over the fabric; the headline is the panel kernel the engine picks",
                    "csr", c2_rows, tkey=f"csr_n{n}_k{k}_band0_ncol{ncol}_vector", x_vec=vx)

                def c5_shard(budget_mb=None):
                    M = ctx.gen_csr_uniform(7 * n, 8 * n, 8 * n, k, band=0, seed=args.seed)
                    if budget_mb and int(M.info.kernel) == 5:
                        M.set_param("twophase_placement_budget_mb", budget_mb)
                        M.set_param("twophase_choose_pieces", 1)
                    if int(M.info.kernel) in (4, 5):
                        M.set_param("panel_keep_csr", 0)
                    return M
                one(f"C5 shard: what the last rank of 8 holds in BASELINE configs[4] (rows {7 * n}-{8 * n} of {8 * n} x {8 * n}, {k} per row; "
                    "x = 640 MB resident); the engine's defaults (piece search within 8 GB)", "csr", c5_shard, tkey=f"csr_n{n}_k{k}_band0_ncol{8 * n}")
                big = args.placement_budget_mb if args.placement_budget_mb != 8192 else 65536
                one(f"C5 shard, piece search of the product stream within {big} MB (what this bench grants its two-phase shards at N > 1: "
                    "--placement-budget-mb)", "csr", lambda: c5_shard(big), tkey=f"csr_n{n}_k{k}_band0_ncol{8 * n}")

            # the partition of a SKEWED matrix.  C4's row-length distribution taken at its quantiles - the rows
            # sorted by length, every heavy row at one end - cut into one shard per GPU (8 shards at N = 1, timed one after the
            # other on the one GPU) by equal rows and by stored entries.  With one GPU per shard the step of the job is its
            # slowest shard: `value` = 2 nnz / that time.  Generated once per rank on its own GPU, partitioned from the
            # device-resident handle (spmv_mat_partition_rows) and cut device to device (spmv_csr_extract_rows).
            def skewed(mode):
                parts = world if world > 1 else 8
                t = time.perf_counter()
                G = ctx.gen_coo_powerlaw(2_000_000, 2_000_000, 4096, seed=args.seed, sorted_by_length=True)
                W = ctx.coo_to_csr(G)
                del G
                bounds = W.partition_rows(parts, mode == "nnz")
                nnz_all = int(W.info.nnz)
                mine = list(range(parts)) if world == 1 else [rank]
                shards = [ctx.extract_rows(W, int(bounds[q]), int(bounds[q + 1])) for q in mine]
                del W
                ctx.sync()
                t_set = time.perf_counter() - t
                vx2 = ctx.gen_vector(2_000_000, seed=args.seed)
                table = torch.zeros(parts, 4, dtype=torch.float64, device=dev)
                for q, S in zip(mine, shards):
                    inf = S.info
                    vy2 = ctx.vector(max(int(inf.nrow), 1))
                    vy2.fill(0.0)
                    for _ in range(3):
                        ctx.apply(S, vx2, vy2)
                    barrier()
                    ms = ctx.apply_timed(S, vx2, vy2, 20)
                    table[q] = torch.tensor([float(inf.nrow), float(inf.nnz), float(int(inf.kernel)), ms], dtype=torch.float64, device=dev)
                    del vy2
                if grouped:
                    dist.all_reduce(table, op=dist.ReduceOp.SUM)  # (one row per rank: a one-hot sum)
                rows_ = table.tolist()
                slowest = max(r_[3] for r_ in rows_)
                most = max(r_[1] for r_ in rows_)
                extra.append({
                    "name": f"C4's row lengths SORTED BY LENGTH (N = 2M, up to 4096 per row, the heavy rows first), {parts} row shards cut by "
                            + ("stored entries (spmv_partition_rows_balanced)" if mode == "nnz" else "equal rows (the reference's split, src/mat_vec.cpp:245-246)"),
                    "format": "csr", "partition": "entries" if mode == "nnz" else "rows", "shards": parts, "nnz": nnz_all,
                    "measured": ("one shard per rank, all ranks at once" if world > 1 else f"the {parts} shards one after the other on ONE GPU"),
                    "entries_per_shard_max_over_mean": round(most * parts / nnz_all, 4),
                    "slowest_shard_ms": round(slowest, 5), "sum_of_shards_ms": round(sum(r_[3] for r_ in rows_), 5),
                    "value": round(2.0 * nnz_all / slowest / 1e6, 2), "unit": "GFLOP/s with one GPU per shard (the step of the job is its slowest shard)",
                    "setup_seconds": round(t_set, 3),
                    "per_shard": [{"rows": int(r_[0]), "entries": int(r_[1]), "kernel_id": int(r_[2]), "ms": round(r_[3], 5)} for r_ in rows_],
                })
                del shards, vx2

            for mode in (("rows", "nnz") if args.partition == "both" else (args.partition,)):
                skewed(mode)

        # every rank's own kernel time (the headline takes the slowest): shows whether one rank's placement / layout lags
        per_rank_ms = [round(kernel_ms, 5)]
        if grouped:
            everyone = torch.zeros(world, dtype=torch.float64, device=dev)
            everyone[rank] = kernel_ms
            dist.all_reduce(everyone, op=dist.ReduceOp.SUM)  # (a one-hot sum: the same collective the times already use)
            per_rank_ms = [round(float(v), 5) for v in everyone.tolist()]
        twophase = None
        if int(info.kernel) == 5:
            twophase = {"panel_cols": A.get_param("twophase_panel_cols"), "padded_entries": A.get_param("twophase_padded"),
                        "product_stream_pieces": A.get_param("twophase_pieces"), "placement_budget_mb": A.get_param("twophase_placement_budget_mb"),
                        "configurations_timed": A.get_param("twophase_placements_timed"),
                        "pieces_exchanged": A.get_param("twophase_pieces_exchanged"),
                        "as_built_over_kept": A.get_param("twophase_placement_spread") / 1000.0}
        # every rank's kernel choice, set-up time and (two-phase) placement search, so that a rank that lags is visible in
        # rank 0's line: one all-gather of five numbers per rank
        mine = [float(int(info.kernel)), setup_s, float(A.get_param("twophase_placements_timed")) if int(info.kernel) == 5 else 0.0,
                A.get_param("twophase_placement_spread") / 1000.0 if int(info.kernel) == 5 else 0.0, kernel_ms]
        per_rank_layout = [mine]
        if grouped:
            table = torch.zeros(world, len(mine), dtype=torch.float64, device=dev)
            table[rank] = torch.tensor(mine, dtype=torch.float64, device=dev)
            dist.all_reduce(table, op=dist.ReduceOp.SUM)
            per_rank_layout = table.tolist()
        per_rank_layout = [{"rank": r, "kernel_id": int(v[0]), "setup_seconds": round(v[1], 3), "twophase_configurations_timed": int(v[2]),
                            "twophase_as_built_over_kept": round(v[3], 3), "kernel_ms": round(v[4], 5)}
                           for r, v in enumerate(per_rank_layout)]
        wall_s, kernel_ms, exch_max = max_over_ranks(wall_s, kernel_ms, exch_s or 0.0)

    if rank == 0:
        nnz_rank = int(info.nnz)
        nnz_total = nnz_rank * world
        gflops = 2.0 * nnz_total * args.steps / wall_s / 1e9
        bytes_launch = algorithmic_bytes("csr", n, ncol, nnz_rank)
        achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9
        kernel_names = {1: "csr_vector_kernel", 2: "csr_ldswin_kernel", 3: "csr_scalar_kernel", 4: "csr_panel_pp_kernel",
                        5: "tp_expand_kernel + tp_reduce_kernel (two-phase)"}
        kernel_name = kernel_names.get(int(info.kernel), str(info.kernel))
        panel = panel_of(A) if int(info.kernel) == 4 else None
        counters = measured_counters(f"csr_n{n}_k{k}_band{args.band}_ncol{ncol}", kernel_name, panel)
        traffic_source = "profiles/pmc_traffic.json (stamped constants of an earlier run)" if counters["traffic"] else None
        if live and live.get("traffic"):
            # measured minutes ago by this command; kept only if the child ran the instance this process ran
            same = live["child"]["kernel_id"] == int(info.kernel) and (panel is None or all(
                int(panel.get(name, -1)) == int(val) for name, val in (live["child"]["panel_layout"] or {}).items()))
            if same:
                counters = dict(counters, traffic=live["traffic"], measured_on=live["kernel"])
                traffic_source = live["source"]
            else:
                live["error"] = f"the counter passes ran {live['child']}, this process {int(info.kernel)} / {panel}: not used"
        l2 = None
        if counters["l2_line_ops"]:
            # what bounds the panel kernel on scattered columns is the L2's line rate, not HBM: 128-byte line
            # operations per launch (L1->L2 read requests + fills, PMC) over this run's kernel time, against the L2s' peak
            l2_gbs = counters["l2_line_ops"] * 128 / (kernel_ms * 1e-3) / 1e9
            l2 = {"l2_line_ops": counters["l2_line_ops"], "l2_achieved": round(l2_gbs, 1), "l2_peak": L2_PEAK_GBS, "l2_unit": "GB/s",
                  "l2_frac": round(l2_gbs / L2_PEAK_GBS, 4)}
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
                "process_group": (f"{backend} (torch.distributed, world {world})" if grouped else "none (plain single-process run)"),
                "kernel": kernel_name,
                "lanes_per_row": int(info.lanes_per_row),
                "setup_seconds": round(setup_s, 3),
                "setup_seconds_without_trials": round(setup_no_trial_s, 3) if setup_no_trial_s is not None else None,
                "setup_seconds_with_plan": round(setup_with_plan_s, 3) if setup_with_plan_s is not None else None,
                "plan": {"bytes": len(my_plan), "shared_from_rank_0": bool(shared_plan), "plans_equal": plans_equal,
                         "note": "N > 1: rank 0's plan (spmv_mat_get_plan) is broadcast and the other ranks build their shards under it "
                                 "(spmv_ctx_set_plan), so same-shape shards cannot end on different kernels; the two-phase piece search "
                                 "stays per rank (physical placement is no part of a plan)"},
                "device_bytes": {"held_during_the_timed_loop": int(bytes_held), "with_the_csr_copy": int(bytes_with_csr),
                                 "matrix_csr": 12 * nnz_rank + 4 * (n + 1),
                                 "ratio_to_matrix": round(bytes_held / (12 * nnz_rank + 4 * (n + 1)), 3)},
                "panel_layout": panel,
                "twophase_layout": twophase,
                "kernel_ms_per_rank": per_rank_ms,
                "per_rank": per_rank_layout,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": counters["traffic"],
                # the counters' bytes over THIS run's kernel time: north_star's "FETCH_SIZE/WRITE_SIZE counters reported as
                # achieved HBM GB/s" (the L2s' traffic with memory, what the chip's ~8 TB/s bound)
                "traffic_gbs": round(counters["traffic"] / (kernel_ms * 1e-3) / 1e9, 1) if counters["traffic"] else None,
                "traffic_frac": round(counters["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if counters["traffic"] else None,
                "traffic_measured_on": counters["measured_on"],
                "traffic_source": traffic_source,
                "traffic_live": ({k_: live.get(k_) for k_ in ("fetch_size_kb", "write_size_kb", "seconds", "error")} if live else None),
                "algorithmic_bytes_per_launch": bytes_launch,
                "kernel_ms": round(kernel_ms, 5),
                **(l2 or {}),
                **({"limiter": "the L2s' 128-byte line rate (one line operation per gathered x line: ~0.63 per entry with 20000 fp64 "
                               "accumulators per CU), not HBM: see l2_frac; the 60 % target is not met on uniform-random columns, and twice "
                               "the accumulators per CU (register file, profiles/r06_probe_regacc.txt) do not get there either (DESIGN.md 4.2)"}
                   if l2 and l2["l2_frac"] > 0.75 else {}),
            },
        }
        if grouped:
            out["with_x_allgather_each_step"] = {
                "value": round(2.0 * nnz_total * args.steps / exch_max / 1e9, 3),
                "unit": "GFLOP/s",
                "allgather_ms": round(allgather_ms, 4) if allgather_ms else None,
                "bytes_per_rank": 8 * n,
            }
            out["y_concatenate_ms"] = round(y_concat_ms, 4) if y_concat_ms is not None else None
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)

    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
