"""CPU tests of bench.py's bookkeeping: the bytes a kernel has to move (`roofline.bytes_required` of the extra lines) and
the stamp check that keeps stale PMC constants out of the line (`roofline.traffic`, `l2_line_ops`)."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def test_required_bytes_per_kernel_kind():
    n, k = 4_000_000, 64
    # C3: ELL's algorithmic figure is the column-reading kernel's requirement; diagonal slots drop the 4-byte index
    assert bench.algorithmic_bytes("ell", n, n, n * k, k) == 3_168_000_000 == bench.required_bytes("ell_columns", n, n, n * k, k)
    assert bench.required_bytes("ell_diagonals", n, n, n * k, k) == 8 * n * k + 8 * n + 16 * n == 2_144_000_000
    # C2: CSR
    assert bench.required_bytes("csr", 10_000_000, 10_000_000, 320_000_000) == 4_120_000_004 == bench.algorithmic_bytes("csr", 10_000_000, 10_000_000, 320_000_000)
    # C4: the panel path moves 12 bytes per entry, the segmented scan COO's 16
    nnz = 115_008_628
    assert bench.required_bytes("coo_segscan", 2_000_000, 2_000_000, nnz) == bench.algorithmic_bytes("coo", 2_000_000, 2_000_000, nnz)
    assert bench.required_bytes("panel", 2_000_000, 2_000_000, nnz) < bench.algorithmic_bytes("coo", 2_000_000, 2_000_000, nnz)


def test_counters_are_reported_only_for_the_kernel_and_layout_they_were_measured_on():
    table = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
    key = "csr_n10000000_k32_band0_ncol10000000"
    layout = dict(table[key]["match"]["panel_layout"])
    got = bench.measured_counters(key, "csr_panel_pp_kernel", layout)
    assert got["traffic"] == table[key]["hbm_bytes_per_launch"] and got["l2_line_ops"] == table[key]["tcp_tcc_read_req"] + table[key]["tcc_miss"]
    assert 270e6 < got["l2_line_ops"] < 280e6
    other = dict(layout, unroll=4)
    stale = bench.measured_counters(key, "csr_panel_pp_kernel", other)
    assert stale["traffic"] is None and stale["l2_line_ops"] is None and "stale" in stale["measured_on"]
    assert bench.measured_counters(key, "tp_expand_kernel + tp_reduce_kernel (two-phase)", None)["traffic"] is None
    assert bench.measured_counters("no_such_workload", "csr_panel_pp_kernel", layout) == {"traffic": None, "l2_line_ops": None, "measured_on": None}
    # every entry of the table that carries a stamp names a kernel
    for name, e in table.items():
        if isinstance(e, dict) and "match" in e:
            assert e["match"].get("kernel"), name


def test_live_counter_rows_are_reduced_to_the_products_of_the_right_kernel():
    """bench.py's own --pmc passes: of a counter_collection.csv only the last three PRODUCT launches of the kernel that ran count
    (trial launches of the panel kernel carry `true` as their fourth template argument; set-up kernels are other kernels)"""
    ns = "spmv::(anonymous namespace)::"
    sig = "(int const*, int, int const*)"
    rows = []
    did = 0
    for name, val in ([(ns + "gen_csr_uniform_kernel(long)", 9e9)] + [(ns + "csr_panel_pp_kernel<8, 4, 2, true, 1>" + sig, 1e9)] * 8 +
                      [(ns + "csr_panel_pp_kernel<8, 4, 2, false, 1>" + sig, v) for v in (7.0, 100.0, 101.0, 102.0)]):
        did += 1
        rows.append({"Kernel_Name": name, "Dispatch_Id": str(did), "Counter_Value": str(val), "Counter_Name": "FETCH_SIZE"})
    per, shown = bench.pmc_mean_of_products(rows, 4)
    assert per == {"csr_panel_pp_kernel": 101.0} and shown == "csr_panel_pp_kernel<8, 4, 2, false, 1>"
    # three product launches that are not one kernel (a handle re-selected between them) are not a measurement
    mixed = rows + [{"Kernel_Name": ns + "csr_panel_pp_kernel<4, 4, 2, false, 3>" + sig, "Dispatch_Id": str(did + 1), "Counter_Value": "5", "Counter_Name": "FETCH_SIZE"}]
    assert "not one kernel" in bench.pmc_mean_of_products(mixed, 4)
    # the two-phase product is two kernels; their means add up in the caller
    tp = [{"Kernel_Name": ns + n + "(int, spmv::(anonymous namespace)::tp_piece_tab, int)", "Dispatch_Id": str(i), "Counter_Value": str(v)}
          for i, (n, v) in enumerate([("tp_expand_kernel<1024, 3, true>", 5.0), ("tp_reduce_kernel", 1.0)] * 4)]
    per, shown = bench.pmc_mean_of_products(tp, 5)
    assert per == {"tp_expand_kernel": 5.0, "tp_reduce_kernel": 1.0} and shown == "tp_reduce_kernel"
    assert "fewer than 3" in bench.pmc_mean_of_products(rows[:10], 4)
    assert "no product kernel" in bench.pmc_mean_of_products(rows, 99)
