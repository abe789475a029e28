"""The drop-in layer end to end on the GPU: the engine's harness with --verify, and — where it was built (the
build container has the reference sources; the binary travels) — the REFERENCE's own main.cpp compiled unchanged
against include/compat, run on a small Matrix Market file."""
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases
from conftest import perf_expect
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
    for name in ("COO", "CSR", "CSR EDIT-IN-PLACE", r"CSR \(edit undone\)", "CSR NUMA", "CSC", "CSC NUMA", "ELL", "ELL NUMA", "COO NUMA"):
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


def test_sharded_drivers_partition_by_rows_or_by_entries(tmp_path, pkg):
    """SURVEY 8e / 8f-4 through the harness: power-law rows sorted by length (the heavy rows at one end), 8 shards.
    `--partition rows` is the reference's split (src/mat_vec.cpp:245-246) and the default; `--partition nnz` (or
    SPMV_COMPAT_PARTITION=nnz for the reference's own main.cpp on the shim) cuts by stored entries.  Every driver verifies
    under both; the printed balance is lopsided for rows and even for entries.  Then the device-side driver (`--sharded`):
    one matrix generated on GPU 0, partitioned there and handed out device to device."""
    import json
    import os

    def balance(out, fmt):
        m = re.search(rf"### {fmt} NUMA shards = 8, partition by (\w+): stored entries per shard max / mean = ([0-9.]+); slowest shard ([0-9.]+) ms", out)
        assert m, out
        return m.group(1), float(m.group(2)), float(m.group(3))

    base = [str(BIN / "spmv_main"), "--synthetic", "powerlaw", "--n", "60000", "--max-len", "1024", "--sorted-by-length", "8",
            "--format", "coo,csr,csc,ell", "--verify", "--reps", "3", "--no-dropin"]
    outs = {}
    for mode, extra_args, env in (("rows", [], None), ("nnz", ["--partition", "nnz"], None)):
        r = subprocess.run(base + extra_args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        for name in ("CSR NUMA", "CSC NUMA", "ELL NUMA", "COO NUMA"):
            m = re.search(rf"### {name} VERIFY .* = ([0-9.e+-]+) OK", r.stdout)
            assert m and float(m.group(1)) <= 1e-10, (mode, name, r.stdout)
        outs[mode] = r.stdout
    for fmt in ("CSR", "COO"):
        how_r, imb_r, _ = balance(outs["rows"], fmt)
        how_n, imb_n, _ = balance(outs["nnz"], fmt)
        assert how_r == "rows" and how_n == "entries"
        assert imb_r > 2.5 and imb_n <= 1.05, (fmt, imb_r, imb_n)
    # ELL stores max_len slots for every row: equal rows are equal work, whatever was asked for
    assert balance(outs["nnz"], "ELL")[1] <= 1.001
    # the device-side driver, both partitions, rows checked against the oracle from the seed
    import oracle_lib as ol

    orc = ol.load_oracle()
    n, max_len = 300_000, 2048
    hr, hc, hv = pkg.synth.coo_powerlaw(n, n, max_len, seed=1, sorted_by_length=True)
    rp = np.concatenate(([0], np.cumsum(np.bincount(hr, minlength=n)))).astype(np.int32)
    x = pkg.synth.vec_uniform(n, seed=1)
    ref, scale = np.zeros(n), np.zeros(n)
    ol.csr_spmv(orc, rp, hc, hv, x, ref)
    ol.csr_abs_row_sums(orc, rp, hc, hv, x, scale)
    seen = {}
    for mode in ("rows", "nnz"):
        chk = tmp_path / f"rows_{mode}.txt"
        r = subprocess.run([str(BIN / "spmv_main"), "--synthetic", "powerlaw", "--n", str(n), "--max-len", str(max_len), "--sorted-by-length",
                            "--sharded", "--gpus", "8", "--partition", mode, "--reps", "3", "--check-rows", "40", "--check-out", str(chk)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        part = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{") and "sharded_partition" in ln][-1]
        assert part["sharded_partition"] == ("entries" if mode == "nnz" else "rows") and len(part["shards"]) == 8
        assert sum(sh["entries"] for sh in part["shards"]) == int(rp[-1]) and sum(sh["rows"] for sh in part["shards"]) == n
        seen[mode] = part
        got = np.array([[float(int(a)), float.fromhex(b)] for a, b in (ln.split() for ln in chk.read_text().splitlines())])
        idx = got[:, 0].astype(np.int64)
        assert len(idx) >= 8 * 3 * 8
        ol.assert_parity(got[:, 1], ref[idx], scale[idx], f"spmv_main --sharded powerlaw, partition {mode}")
    assert seen["rows"]["entries_per_shard_max_over_mean"] > 2.5 and seen["nnz"]["entries_per_shard_max_over_mean"] <= 1.02
    # with one GPU per shard the job's step is its slowest shard: the even split's slowest shard is the faster one
    perf_expect(seen["nnz"]["slowest_shard_ms"] < seen["rows"]["slowest_shard_ms"], f"slowest shard: by entries {seen['nnz']['slowest_shard_ms']} ms, by rows {seen['rows']['slowest_shard_ms']} ms")


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
    # round 6: the reader and the converting constructors upload their container at their end (spmv_compat_prefetch), so a format's
    # first timed product no longer carries context creation, upload and kernel selection: COO (main.cpp's first loop) and CSC
    # used to print 0.1 and 2-5 GFLOP/s on C1 for that reason.  A wall-clock statement: recorded, asserted under -m gpu_perf
    rate = {fmt: float(re.search(rf"### {fmt} CPU GFLOPS = ([0-9.]+)", r.stdout).group(1)) for fmt in ("COO", "CSR", "CSC", "ELL")}
    perf_expect(min(rate.values()) > 0.25 * max(rate.values()), f"ref_main: one format's timed loop carries its set-up: {rate}")
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

    # bench.py sets HSA_ENABLE_IPC_MODE_LEGACY=0 itself (dmabuf IPC: what RCCL's cross-process buffers need on this platform)
    # before anything touches the GPU: the test takes the variable out of the environment to show that nothing depends on
    # the caller having exported it
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--rows", "400000", "--steps", "4", "--warmup", "1",
                        "--cpu-sample-rows", "100000", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["steps"] == 4 and line["unit"] == "GFLOP/s" and line["value"] > 0
    assert line["dtype"] == "f64" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1.0
    assert line["config"]["process_group"].startswith("none")
    names = " | ".join(e["name"] for e in line["extra"])
    assert "configs[2] names" in names and "configs[3] names" in names and "band of 65536" in names
    skew = {e["partition"]: e for e in line["extra"] if "partition" in e}  # the skewed matrix cut by equal rows and by entries
    assert set(skew) == {"rows", "entries"} and all(e["shards"] == 8 and len(e["per_shard"]) == 8 for e in skew.values())
    assert skew["rows"]["entries_per_shard_max_over_mean"] > 3.0 and skew["entries"]["entries_per_shard_max_over_mean"] <= 1.02
    perf_expect(skew["entries"]["slowest_shard_ms"] < skew["rows"]["slowest_shard_ms"] and skew["entries"]["value"] > skew["rows"]["value"],
                f"bench.py skewed shards: by entries {skew['entries']['slowest_shard_ms']} ms, by rows {skew['rows']['slowest_shard_ms']} ms")
    for e in line["extra"]:  # a fraction of the bytes the kernel has to move can never exceed 1
        if "partition" in e:
            continue
        assert 0 < e["roofline"]["frac"] <= 1.0 and e["roofline"]["bytes_required"] > 0, e
    kernels = {e["name"]: e["kernel"] for e in line["extra"] if "kernel" in e}
    assert any("ell_kernel_x2" in v for v in kernels.values()) and any("coo_segscan_kernel" in v for v in kernels.values())
    assert any("coo_segscan_bins_kernel" in v for v in kernels.values())  # the scan over the copy in column bins, one per XCD
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["value"] > 0
    # counter traffic as a rate (north_star: "counters reported as achieved HBM GB/s"): live passes or the stamped constants or null
    rl = line["roofline"]
    assert "traffic_gbs" in rl and "traffic_source" in rl and (rl["traffic"] is None or abs(rl["traffic_gbs"] - rl["traffic"] / rl["kernel_ms"] / 1e6) < 1.0)
    assert len(line["config"]["per_rank"]) == 1 and line["config"]["per_rank"][0]["kernel_ms"] > 0 and "y_concatenate_ms" not in line
    assert line["config"]["plan"]["plans_equal"] is None and line["config"]["plan"]["shared_from_rank_0"] is False  # (no process group: nothing to share)
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
    # the optional gather of the y slices is timed by itself, and every rank's kernel / layout facts reach rank 0's line
    assert line["y_concatenate_ms"] > 0
    ranks = line["config"]["per_rank"]
    assert [r_["rank"] for r_ in ranks] == [0, 1] and all(r_["kernel_ms"] > 0 and r_["kernel_id"] in (1, 2, 4, 5) for r_ in ranks)
    assert line["config"]["kernel_ms_per_rank"] == [r_["kernel_ms"] for r_ in ranks]
    # rank 0's plan was broadcast and rank 1 built its shard under it: the two cannot have drawn different kernels or parameters
    plan = line["config"]["plan"]
    assert plan["shared_from_rank_0"] is True and plan["plans_equal"] is True and plan["bytes"] >= 16 + 128
    assert ranks[0]["kernel_id"] == ranks[1]["kernel_id"]
    skew = {e["partition"]: e for e in line["extra"] if "partition" in e}  # one shard per rank, both partitions, every rank's time on rank 0's line
    assert set(skew) == {"rows", "entries"} and all(e["shards"] == 2 and all(sh["ms"] > 0 for sh in e["per_shard"]) for e in skew.values())
    assert skew["rows"]["entries_per_shard_max_over_mean"] > 1.5 and skew["entries"]["entries_per_shard_max_over_mean"] <= 1.02
    band = [e for e in line["extra"] if "band of 65536" in e["name"]]  # the band-random variant is reported at every world size
    assert len(band) == 1 and band[0]["n_gpus"] == 2 and band[0]["nnz"] == 2 * 400000 * 32 and 0 < band[0]["roofline"]["frac"] <= 1.0


def _check_sharded_rows(pkg, orc, path, n, parts, k, band, seed):
    """y rows written by `spmv_main --sharded --check-rows` against the oracle on the same rows, regenerated from the seed"""
    import oracle_lib as ol

    ncol = n * parts
    x = pkg.synth.vec_uniform(ncol, seed=seed)
    rows, vals = [], []
    for ln in Path(path).read_text().splitlines():
        r, v = ln.split()
        rows.append(int(r))
        vals.append(float.fromhex(v))
    rows, vals = np.array(rows), np.array(vals)
    assert len(rows) > 0
    worst = 0.0
    # the file holds runs of consecutive rows (first / middle / last of every shard)
    cuts = np.flatnonzero(np.diff(rows) != 1) + 1
    for seg_r, seg_v in zip(np.split(rows, cuts), np.split(vals, cuts)):
        r0, r1 = int(seg_r[0]), int(seg_r[-1]) + 1
        rp, col, val = pkg.synth.csr_uniform(r0, r1, ncol, k, band=band, seed=seed)
        ref, scale = np.zeros(r1 - r0), np.zeros(r1 - r0)
        ol.csr_spmv(orc, rp, col, val, x, ref, fma=True)
        ol.csr_abs_row_sums(orc, rp, col, val, x, scale)
        worst = max(worst, float(np.max(np.abs(seg_v - ref) / np.maximum(scale, 1e-300))))
    assert worst <= 1e-10, worst
    return len(rows)


@pytest.mark.parametrize("parts,n,k,band", [(3, 50_000, 12, 0), (8, 250_000, 32, 0), (4, 300_000, 32, 4096)])
def test_native_sharded_harness_on_device_generated_shards(tmp_path, pkg, orc, parts, n, k, band):
    """`spmv_main --synthetic ... --sharded --gpus P`: the C++ driver of BASELINE configs[4] - shards generated on their
    devices, x slices all-gathered by spmv_comm_*, the reference's timed loop and print-out, one JSON line - with P
    participants on however many GPUs the box has.  Sampled rows of every shard are checked against the oracle."""
    import json

    out = tmp_path / "rows.txt"
    r = subprocess.run([str(BIN / "spmv_main"), "--synthetic", "band" if band else "uniform", "--n", str(n), "--k", str(k), "--band", str(band),
                        "--seed", "5", "--sharded", "--gpus", str(parts), "--reps", "10", "--check-rows", "700", "--check-out", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(rf"### ROW={n * parts}, COL={n * parts}, NNZ={n * parts * k} ", r.stdout), r.stdout
    assert re.search(r"### CSR NUMA GFLOPS = [0-9.]+", r.stdout)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["participants"] == parts and line["nnz_total"] == n * parts * k and line["exchange"] in ("rccl", "peer-copy")
    assert line["gflops"] > 0 and line["with_x_allgather_each_step"]["gflops"] > 0
    # same-shape shards are built under shard 0's plan: none of them can have drawn another kernel or another parameter
    assert re.search(r"### CSR NUMA shards built under shard 0's plan \(\d+ bytes\): plans equal = yes", r.stdout), r.stdout[-1500:]
    assert _check_sharded_rows(pkg, orc, out, n, parts, k, band, 5) == 3 * 700 * parts


def test_native_sharded_harness_with_two_phase_shards_and_a_granted_search_budget(tmp_path, pkg, orc):
    """twelve shards of 2.5M rows x 30M columns (x twelve times as long as a shard has rows: the two-phase kernel, each stream
    0.66 GB = one piece, searched): `--placement-budget-mb` grants the piece search of every shard its budget after the
    build; rows of every shard against the oracle, and the queued variant printed beside the reference-protocol line"""
    import json

    out = tmp_path / "rows.txt"
    parts, n, k = 12, 2_500_000, 32
    r = subprocess.run([str(BIN / "spmv_main"), "--synthetic", "uniform", "--n", str(n), "--k", str(k), "--seed", "5", "--sharded", "--gpus", str(parts),
                        "--reps", "4", "--placement-budget-mb", "3072", "--check-rows", "400", "--check-out", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert re.search(r"### CSR NUMA GFLOPS = [0-9.]+", r.stdout) and re.search(r"### CSR NUMA GFLOPS, all repetitions queued and one wait = [0-9.]+", r.stdout)
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["participants"] == parts and line["kernel_of_shard_0"] == 5 and line["queued_one_wait"]["gflops"] > 0
    assert _check_sharded_rows(pkg, orc, out, n, parts, k, 0, 5) == 3 * 400 * parts


def test_native_sharded_harness_at_the_full_shard_shape_of_config_5(tmp_path, pkg, orc):
    """two of the eight shards of BASELINE configs[4] as the native driver builds them is what one GPU's time allows here:
    `--n 10000000 --gpus 2` = shards of 10M rows x 20M columns, 6.4e8 entries in all, no host container anywhere"""
    import json

    out = tmp_path / "rows.txt"
    r = subprocess.run([str(BIN / "spmv_main"), "--synthetic", "uniform", "--n", "10000000", "--k", "32", "--seed", "1", "--sharded", "--gpus", "2",
                        "--reps", "10", "--check-rows", "1000", "--check-out", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["nnz_total"] == 640_000_000 and line["ncol"] == 20_000_000
    assert _check_sharded_rows(pkg, orc, out, 10_000_000, 2, 32, 0, 1) == 6000


def test_bench_under_torchrun_with_one_rank_runs_the_rccl_collectives(tmp_path):
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launcher starts before anything touches the GPU,
    the process group is backend nccl (= RCCL) with world size 1, and the run makes the same collective calls as the
    8-GPU job - all_gather_into_tensor of the x slices, barriers, the all-reduce of the times."""
    import json
    import os
    import socket
    import sys

    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # bench.py sets it itself
    env.pop("SPMV_BENCH_BACKEND", None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "1", "--rows", "400000",
                        "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["config"]["process_group"].startswith("nccl")
    assert line["with_x_allgather_each_step"]["value"] > 0 and line["with_x_allgather_each_step"]["allgather_ms"] > 0
    assert line["y_concatenate_ms"] > 0 and len(line["config"]["per_rank"]) == 1


def test_allgather_x_over_nccl_world_one():
    """arm-spmv_amd/dist.py on backend nccl with one rank, in a child process (its process group must not leak into the
    test session): the equal-slice all-gather, concatenate_y, the max / sum reductions of the timing and solver paths"""
    import os
    import socket
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = f"""
import sys, torch, torch.distributed as dist
sys.path.insert(0, {str(ROOT)!r})
from __graft_entry__ import load_package
load_package()
import importlib
shard = importlib.import_module("arm_spmv_amd.dist")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
n = 1_000_003
own = torch.rand(n, dtype=torch.float64, device=dev)
full = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
shard.allgather_x(full, own, n)
torch.cuda.synchronize()
assert torch.equal(full, own)
assert torch.equal(shard.concatenate_y(own, n), own)
assert shard.max_over_ranks(3.5, dev) == 3.5 and shard.sum_over_ranks(2.25, dev) == 2.25
dist.barrier()
dist.destroy_process_group()
print("nccl world 1 OK")
"""
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)  # arm-spmv_amd/dist.py sets it when imported
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "nccl world 1 OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


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
