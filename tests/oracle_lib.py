"""ctypes access to the oracle — TEST INFRASTRUCTURE ONLY.

Two libraries:
  oracle/_build/libspmv_oracle.so   our plain-C restatement of the reference loops (oracle/spmv_oracle.c)
  oracle/_ref/libarmspmv_ref.so     the real reference, compiled from /root/reference by oracle/Makefile
                                    (present only where it was built: the build container; the prebuilt
                                    file travels to the GPU box with the snapshot)
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "_build" / "libspmv_oracle.so"
REF_SO = ROOT / "oracle" / "_ref" / "libarmspmv_ref.so"

_vp = C.c_void_p


def _p(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def load_oracle() -> C.CDLL:
    override = os.environ.get("SPMV_ORACLE_SO")  # e.g. an AddressSanitizer build of the same sources (CPU only)
    if override:
        lib = C.CDLL(override)
    else:
        if not ORACLE_SO.exists():
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
        lib = C.CDLL(str(ORACLE_SO))
    lib.orc_dot.restype = C.c_double
    lib.orc_dot_fma.restype = C.c_double
    for name in ("orc_coo_to_csr", "orc_coo_max_row_nnz", "orc_csr_count_diags"):
        getattr(lib, name).restype = C.c_int32
    return lib


def ref_available() -> bool:
    return REF_SO.exists()


def load_ref() -> C.CDLL:
    lib = C.CDLL(str(REF_SO))
    lib.ref_dot.restype = C.c_double
    lib.ref_coo_to_ell.restype = C.c_int
    lib.ref_csr_to_dia.restype = C.c_int
    return lib


# ---------------------------------------------------------------- oracle wrappers (y updated in place)
def _sfx(fma: bool) -> str:
    return "_fma" if fma else ""


def coo_spmv(lib, row, col, val, x, y, fma=False):
    getattr(lib, "orc_coo_spmv" + _sfx(fma))(C.c_int64(len(val)), _p(row), _p(col), _p(val), _p(x), _p(y))


def csr_spmv(lib, row_ptr, col, val, x, y, fma=False):
    getattr(lib, "orc_csr_spmv" + _sfx(fma))(C.c_int32(len(row_ptr) - 1), _p(row_ptr), _p(col), _p(val), _p(x), _p(y))


def symgs(lib, row_ptr, col, val, b, x, sweeps=1, order=None):
    """x updated in place, rows swept in `order` (None: 0..n-1); returns 0 or 1 + the first row without a diagonal"""
    lib.orc_symgs_ordered.restype = C.c_int32
    return lib.orc_symgs_ordered(C.c_int32(len(row_ptr) - 1), _p(row_ptr), _p(col), _p(val), _p(b), _p(x), C.c_int32(sweeps),
                                 _p(i32(order)) if order is not None else None)


def greedy_colour_order(lib, row_ptr, col):
    """(number of colours, colour[], order[]) of the sequential greedy colouring in row order"""
    n = len(row_ptr) - 1
    colour, order = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    lib.orc_greedy_colour_order.restype = C.c_int32
    return lib.orc_greedy_colour_order(C.c_int32(n), _p(row_ptr), _p(col), _p(colour), _p(order)), colour, order


def csr_spmv_omp(lib, row_ptr, col, val, x, y):
    lib.orc_csr_spmv_omp(C.c_int32(len(row_ptr) - 1), _p(row_ptr), _p(col), _p(val), _p(x), _p(y))


def csc_spmv(lib, col_ptr, row, val, x, y, fma=False):
    getattr(lib, "orc_csc_spmv" + _sfx(fma))(C.c_int32(len(col_ptr) - 1), _p(col_ptr), _p(row), _p(val), _p(x), _p(y))


def ell_spmv(lib, nrow, k, col, val, x, y, fma=False):
    getattr(lib, "orc_ell_spmv" + _sfx(fma))(C.c_int32(nrow), C.c_int32(k), _p(col), _p(val), _p(x), _p(y))


def dia_spmv(lib, nrow, offsets, val, x, y, fma=False):
    getattr(lib, "orc_dia_spmv" + _sfx(fma))(C.c_int32(nrow), C.c_int32(len(offsets)), _p(offsets), _p(val), _p(x), _p(y))


def coo_to_csr(lib, nrow, row, col, val):
    nnz = len(val)
    row_ptr = np.zeros(nrow + 1, dtype=np.int32)
    oc = np.zeros(nnz, dtype=np.int32)
    ov = np.zeros(nnz, dtype=np.float64)
    lib.orc_coo_to_csr(C.c_int32(nrow), C.c_int64(nnz), _p(row), _p(col), _p(val), _p(row_ptr), _p(oc), _p(ov), None)
    return row_ptr, oc, ov


def coo_to_csc(lib, ncol, row, col, val):
    nnz = len(val)
    col_ptr = np.zeros(ncol + 1, dtype=np.int32)
    orow = np.zeros(nnz, dtype=np.int32)
    ov = np.zeros(nnz, dtype=np.float64)
    lib.orc_coo_to_csc(C.c_int32(ncol), C.c_int64(nnz), _p(row), _p(col), _p(val), _p(col_ptr), _p(orow), _p(ov))
    return col_ptr, orow, ov


def coo_to_ell(lib, nrow, row, col, val):
    nnz = len(val)
    k = int(lib.orc_coo_max_row_nnz(C.c_int32(nrow), C.c_int64(nnz), _p(row)))
    oc = np.zeros(max(nrow * k, 1), dtype=np.int32)
    ov = np.zeros(max(nrow * k, 1), dtype=np.float64)
    lib.orc_coo_to_ell(C.c_int32(nrow), C.c_int32(k), C.c_int64(nnz), _p(row), _p(col), _p(val), _p(oc), _p(ov))
    return k, oc[: nrow * k], ov[: nrow * k]


def csr_to_dia(lib, nrow, ncol, row_ptr, col, val):
    nd = int(lib.orc_csr_count_diags(C.c_int32(nrow), C.c_int32(ncol), _p(row_ptr), _p(col), None))
    offsets = np.zeros(max(nd, 1), dtype=np.int32)
    lib.orc_csr_count_diags(C.c_int32(nrow), C.c_int32(ncol), _p(row_ptr), _p(col), _p(offsets))
    ov = np.zeros(max(nrow * nd, 1), dtype=np.float64)
    lib.orc_csr_to_dia(C.c_int32(nrow), C.c_int32(ncol), _p(row_ptr), _p(col), _p(val), C.c_int32(nd), _p(offsets), _p(ov))
    return offsets[:nd], ov[: nrow * nd]


def partition_rows(lib, nrow, nparts, part):
    b, e = C.c_int64(), C.c_int64()
    lib.orc_partition_rows(C.c_int64(nrow), C.c_int32(nparts), C.c_int32(part), C.byref(b), C.byref(e))
    return b.value, e.value


def csr_shard_row_ptr(lib, row_ptr, begin, end):
    sub = np.zeros(end - begin + 1, dtype=np.int32)
    lib.orc_csr_shard_row_ptr(_p(row_ptr), C.c_int64(begin), C.c_int64(end), _p(sub))
    return sub


def dot(lib, x, y, fma=False):
    return float(getattr(lib, "orc_dot" + _sfx(fma))(C.c_int64(len(x)), _p(x), _p(y)))


def axpby(lib, alpha, x, beta, y, w, fma=False):
    getattr(lib, "orc_axpby" + _sfx(fma))(C.c_int64(len(w)), C.c_double(alpha), _p(x), C.c_double(beta), _p(y), _p(w))


def csr_abs_row_sums(lib, row_ptr, col, val, x, s):
    lib.orc_csr_abs_row_sums(C.c_int32(len(row_ptr) - 1), _p(row_ptr), _p(col), _p(val), _p(x), _p(s))


# ---------------------------------------------------------------- the parity gate
REL_TOL = 1e-10  # BASELINE.json north_star: "y within 1e-10 relative error of the CPU reference"


def parity_errors(got, ref, scale):
    """(norm-wise relative error, max element error scaled by (|A||x|)_i)"""
    got, ref, scale = f64(got), f64(ref), f64(scale)
    denom = float(np.max(np.abs(ref))) if ref.size else 0.0
    normwise = float(np.max(np.abs(got - ref))) / denom if denom > 0 else float(np.max(np.abs(got - ref), initial=0.0))
    safe = np.where(scale > 0, scale, 1.0)
    elem = float(np.max(np.where(scale > 0, np.abs(got - ref) / safe, np.abs(got - ref)), initial=0.0))
    return normwise, elem


def assert_parity(got, ref, scale, what="", reps=1):
    """both gates of SURVEY 8d; `reps` accumulating calls scale |A||x| by reps"""
    normwise, elem = parity_errors(got, ref, np.asarray(scale) * reps)
    assert normwise <= REL_TOL, f"{what}: norm-wise relative error {normwise:.3e} > {REL_TOL}"
    assert elem <= REL_TOL, f"{what}: |dy_i|/(|A||x|)_i = {elem:.3e} > {REL_TOL}"
    return normwise, elem
