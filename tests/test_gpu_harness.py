"""The drop-in layer end to end on the GPU: the engine's harness with --verify, and — where it was built (the
build container has the reference sources; the binary travels) — the REFERENCE's own main.cpp compiled unchanged
against include/compat, run on a small Matrix Market file."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases
from test_host_io import _write_mtx

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "arm-spmv_amd" / "bin"


def _mtx(tmp_path, pkg, n, k, seed):
    rp, col, val = pkg.synth.csr_uniform(0, n, n, k, seed=seed)
    c = dict(nrow=n, ncol=n, row=np.repeat(np.arange(n, dtype=np.int32), k), col=col, val=val)
    p = tmp_path / f"m{n}.mtx"
    _write_mtx(p, c)
    return p


def test_spmv_main_verifies_every_format_and_the_sharded_drivers(tmp_path, pkg):
    p = _mtx(tmp_path, pkg, 3000, 12, 17)
    r = subprocess.run([str(BIN / "spmv_main"), str(p), "8", "--format", "coo,csr,csc,ell,dia", "--verify", "--reps", "50"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert "### ROW=3000, COL=3000, NNZ=36000" in out
    for name in ("CSR", "CSR EDIT-IN-PLACE", r"CSR \(edit undone\)", "CSR NUMA", "CSC", "CSC NUMA", "ELL", "ELL NUMA", "COO NUMA"):
        m = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", out)
        assert m and float(m.group(1)) <= 1e-10, (name, out)
    # the generated matrix has duplicate (i,j) entries in some rows: the reference's DIA keeps the last one
    # (src/matrix.cpp:721), so DIA is reported, not gated
    assert re.search(r"### DIA VERIFY \(informational\)", out)
    assert re.search(r"### CSC NUMA GFLOPS = [0-9.]+", out)
    for name in ("COO", "CSR", "ELL"):
        assert re.search(rf"### {name} NUMA GFLOPS = [0-9.]+", out) and re.search(rf"### {name} GPU-RESIDENT GFLOPS = [0-9.]+", out)
    m = re.search(r"### DIA NUMA VERIFY .* = ([0-9.e+-]+) OK", out)
    assert m and float(m.group(1)) <= 1e-10 and re.search(r"### DIA NUMA GFLOPS = [0-9.]+", out), out


def test_sharded_drivers_assemble_x_through_the_native_exchange(tmp_path, pkg):
    """SPMV_COMPAT_X_PARTS=3: x reaches the GPU in three slices owned by three participants and every replica is
    assembled by spmv_comm_allgather (device-to-device on this one-GPU box, peer copies / RCCL between GPUs)"""
    import os

    p = _mtx(tmp_path, pkg, 2500, 9, 23)
    env = dict(os.environ, SPMV_COMPAT_X_PARTS="3")
    r = subprocess.run([str(BIN / "spmv_main"), str(p), "5", "--format", "coo,csr,ell,dia", "--verify", "--reps", "4"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in ("CSR NUMA", "ELL NUMA", "COO NUMA", "DIA NUMA"):
        m = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", r.stdout)
        assert m and float(m.group(1)) <= 1e-10, (name, r.stdout)


def test_sharded_drivers_with_more_shards_than_rows(tmp_path, pkg):
    """`spmv_main tri8.mtx 64`: 64 shards over 8 rows.  The reference gives every thread nrow / nthreads = 0 rows and the
    last one all of them (src/mat_vec.cpp:233,245-246); every driver must run (empty shards included) and verify."""
    c = cases.tri8()
    p = tmp_path / "tri8.mtx"
    _write_mtx(p, c)
    r = subprocess.run([str(BIN / "spmv_main"), str(p), "64", "--format", "coo,csr,csc,ell,dia", "--verify", "--reps", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for name in ("CSR NUMA", "CSC NUMA", "ELL NUMA", "COO NUMA", "DIA NUMA"):
        m = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", r.stdout)
        assert m and float(m.group(1)) <= 1e-10, (name, r.stdout)


def test_reference_main_cpp_runs_unchanged_on_the_engine(tmp_path, pkg):
    exe = BIN / "ref_main"
    if not exe.exists():
        pytest.skip("ref_main is built only where the reference sources exist (build container)")
    p = _mtx(tmp_path, pkg, 1500, 8, 23)
    r = subprocess.run([str(exe), str(p), "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "### ROW=1500, COL=1500, NNZ=12000" in r.stdout
    for fmt in ("COO", "CSR", "CSC", "ELL", "DIA"):  # main.cpp:61,71,81,91,101 and src/mat_vec.cpp:216,285,354,415,470
        assert re.search(rf"### {fmt} CPU GFLOPS = [0-9.]+", r.stdout), fmt
        assert re.search(rf"### {fmt} NUMA GFLOPS = [0-9.]+", r.stdout), fmt
    # main.cpp:20-24: no arguments -> usage, return -1
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 255 and "Usage" in r.stdout


def test_bench_contract_one_rank_and_two_rank_rehearsal(tmp_path):
    """bench.py end to end on small shards: the single-rank line carries roofline + cpu_baseline; the two-rank run
    (both ranks on this one GPU, gloo standing in for RCCL: SPMV_BENCH_BACKEND=gloo) exercises sharding, the x
    all-gather and the max-over-ranks timing of the N > 1 path.  Parity of the shards themselves is covered by
    test_row_shards_concatenate_to_unsharded_result."""
    import json
    import os
    import socket
    import sys

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--rows", "400000", "--steps", "4", "--warmup", "1",
                        "--cpu-sample-rows", "100000", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["unit"] == "GFLOP/s" and line["value"] > 0
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1.2
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env["SPMV_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--rows", "400000",
                        "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ncol"] == 800000 and line["config"]["nnz_total"] == 2 * 400000 * 32
    assert line["value"] > 0 and "with_x_allgather_each_step" in line


def test_spmv_main_on_a_stencil_matrix_file(tmp_path):
    """A 5-point Laplacian on a 220 x 220 grid through the reference's own flow (Matrix Market file -> COOMatrixRead ->
    CSRMatrix / ELLMatrix / DIAMatrix constructors -> products): the ELL container the reference builds has slots that
    are diagonals for the interior rows (boundary rows are shorter and padded), the DIA one has five offsets within a
    band — both take the kernels that read x through LDS, ELL without its column stream; every product must verify
    against the COO loop, sharded drivers included."""
    m = 220
    n = m * m
    idx = np.arange(n).reshape(m, m)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [np.full(n, 4.0)]
    for a, b in ((idx[:, :-1], idx[:, 1:]), (idx[:-1, :], idx[1:, :])):
        rows += [a.ravel(), b.ravel()]
        cols += [b.ravel(), a.ravel()]
        vals += [np.full(a.size, -1.0)] * 2
    r, c, v = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals) * np.random.default_rng(3).uniform(0.5, 1.5, 5 * n - 4 * m)
    o = np.lexsort((c, r))  # row-sorted, as the reference's COO shard driver wants it
    p = tmp_path / "lap220.mtx"
    _write_mtx(p, dict(nrow=n, ncol=n, row=r[o].astype(np.int32), col=c[o].astype(np.int32), val=v[o]))
    res = subprocess.run([str(BIN / "spmv_main"), str(p), "4", "--format", "coo,csr,csc,ell,dia", "--verify", "--reps", "5"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert f"### ROW={n}, COL={n}, NNZ={len(v)}" in res.stdout and "### DIA ndiags = 5" in res.stdout
    for name in ("CSR", "CSR NUMA", "CSC", "CSC NUMA", "ELL", "ELL NUMA", "COO NUMA", "DIA NUMA"):
        mm = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", res.stdout)
        assert mm and float(mm.group(1)) <= 1e-10, (name, res.stdout)
    mm = re.search(r"### DIA VERIFY \(informational\) .* = ([0-9.e+-]+)", res.stdout)
    assert mm and float(mm.group(1)) <= 1e-10, res.stdout  # no duplicate entries here: DIA agrees as well


@pytest.mark.parametrize("seed", range(8))
def test_spmv_main_on_random_files_shard_counts_and_x_slicings(tmp_path, pkg, seed):
    """the whole drop-in flow (file -> COOMatrixRead -> constructors -> products -> sharded drivers) on random square
    matrices with random row lengths, shard counts from 1 to more than there are rows, and x assembled from 1-5 slices"""
    import os

    rng = np.random.default_rng(500 + seed)
    n = int(rng.choice([1, 2, 37, 800, 6000]))
    lens = rng.integers(0, 12, n)
    row = np.repeat(np.arange(n, dtype=np.int32), lens)
    col = rng.integers(0, n, len(row)).astype(np.int32)
    key = row.astype(np.int64) * n + col
    _, first = np.unique(key, return_index=True)  # no duplicate (i, j): the reference's DIA would keep the last one only
    first.sort()
    row, col = row[first], col[first]
    val = rng.uniform(-1, 1, len(row))
    if len(row) == 0:
        row, col, val = np.zeros(1, np.int32), np.zeros(1, np.int32), np.ones(1)
    p = tmp_path / f"r{seed}.mtx"
    _write_mtx(p, dict(nrow=n, ncol=n, row=row, col=col, val=val))
    shards = int(rng.choice([1, 2, 3, 7, 16, 2 * n + 3]))
    env = dict(os.environ)
    parts = int(rng.choice([0, 2, 5]))
    if parts:
        env["SPMV_COMPAT_X_PARTS"] = str(parts)
    fmts = "coo,csr,csc,ell" + (",dia" if n <= 800 else "")  # (DIA of a random matrix has ~n diagonals)
    r = subprocess.run([str(BIN / "spmv_main"), str(p), str(shards), "--format", fmts, "--verify", "--reps", "3"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (seed, n, shards, parts, r.stdout[-1500:] + r.stderr[-1500:])
    names = ["CSR", "CSR NUMA", "CSC", "CSC NUMA", "ELL", "ELL NUMA", "COO NUMA"] + (["DIA NUMA"] if n <= 800 else [])
    for name in names:
        m = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", r.stdout)
        assert m and float(m.group(1)) <= 1e-10, (seed, n, shards, parts, name, r.stdout[-1500:])
