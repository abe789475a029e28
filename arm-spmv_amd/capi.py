"""ctypes binding of the C ABI in include/spmv_abi.h (libspmv_hip.so).

This is the Python-side twin of the cgo/ctypes stub shown in INTEGRATION.md: it declares every
entry point of the header, turns negative status codes into exceptions and wraps the opaque
handles in small classes.  There is no fallback of any kind: if the shared library is missing or
no GPU is visible, loading / context creation raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SPMV_HIP_SO") or _PKG / "lib" / "libspmv_hip.so")  # override: A/B against another build

FMT_COO, FMT_CSR, FMT_CSC, FMT_ELL, FMT_DIA = 0, 1, 2, 3, 4
CSR_AUTO, CSR_VECTOR, CSR_LDSWIN, CSR_SCALAR, CSR_PANEL, CSR_TWOPHASE, CSR_SEGSCAN, CSR_SPLIT, CSR_ELL = 0, 1, 2, 3, 4, 5, 6, 7, 8
PRECOND_NONE, PRECOND_JACOBI, PRECOND_SYMGS = 0, 1, 2
FLAG_DPP_REDUCE, FLAG_XCD_REMAP = 1, 2

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p


class MatInfo(C.Structure):
    _fields_ = [
        ("format", C.c_int32),
        ("nrow", C.c_int32),
        ("ncol", C.c_int32),
        ("ell_k", C.c_int32),
        ("nnz", C.c_int64),
        ("row_begin", C.c_int64),
        ("max_row_nnz", C.c_int32),
        ("kernel", C.c_int32),
        ("lanes_per_row", C.c_int32),
        ("sorted_rows", C.c_int32),
        ("device_bytes", C.c_int64),
    ]


# name -> (restype, argtypes).  Kept in one table so tests can check it against the header.
SIGNATURES = {
    "spmv_abi_version": (C.c_int, []),
    "spmv_last_error": (C.c_char_p, []),
    "spmv_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "spmv_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "spmv_ctx_create_on_stream": (C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    "spmv_ctx_destroy": (C.c_int, [_vp]),
    "spmv_sync": (C.c_int, [_vp]),
    "spmv_ctx_mem_info": (C.c_int, [_vp, _i64p, _i64p]),
    "spmv_ctx_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "spmv_ctx_get_param": (C.c_int, [_vp, C.c_char_p, _i64p]),
    "spmv_vec_create": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "spmv_vec_wrap_device": (C.c_int, [_vp, C.c_int64, _vp, C.POINTER(_vp)]),
    "spmv_vec_destroy": (C.c_int, [_vp]),
    "spmv_vec_size": (C.c_int, [_vp, _i64p]),
    "spmv_vec_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "spmv_vec_upload": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "spmv_vec_download": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "spmv_vec_fill": (C.c_int, [_vp, C.c_double]),
    "spmv_vec_copy": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64]),
    "spmv_comm_create": (C.c_int, [C.POINTER(_vp), C.c_int32, C.POINTER(_vp)]),
    "spmv_comm_destroy": (None, [_vp]),
    "spmv_comm_backend": (C.c_char_p, [_vp]),
    "spmv_comm_allgather": (C.c_int, [_vp, C.POINTER(_vp), _i64p]),
    "spmv_csr_upload": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_csr_wrap_device": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_csr_upload_shard": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_coo_upload": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_coo_wrap_device": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_ell_upload": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, C.POINTER(_vp)]),
    "spmv_ell_wrap_device": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, _vp, _vp, C.POINTER(_vp)]),
    "spmv_csc_upload": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.POINTER(_vp)]),
    "spmv_dia_upload": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.POINTER(_vp)]),
    "spmv_mat_destroy": (C.c_int, [_vp]),
    "spmv_mat_get_info": (C.c_int, [_vp, C.POINTER(MatInfo)]),
    "spmv_mat_validate": (C.c_int, [_vp]),
    "spmv_mat_set_kernel": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "spmv_mat_set_flags": (C.c_int, [_vp, C.c_uint32]),
    "spmv_mat_set_param": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "spmv_mat_get_param": (C.c_int, [_vp, C.c_char_p, _i64p]),
    "spmv_mat_get_plan": (C.c_int, [_vp, _vp, _i64p]),
    "spmv_mat_set_plan": (C.c_int, [_vp, _vp, C.c_int64]),
    "spmv_ctx_set_plan": (C.c_int, [_vp, _vp, C.c_int64]),
    "spmv_plan_check": (C.c_int, [_vp, C.c_int64, _i32p, _i32p, _i32p]),
    "spmv_mat_download": (C.c_int, [_vp, _vp, _vp, _vp]),
    "spmv_mat_device_ptrs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "spmv_apply": (C.c_int, [_vp, _vp, _vp, _vp]),
    "spmv_apply_timed": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _f64p]),
    "spmv_dot": (C.c_int, [_vp, _vp, _vp, _f64p]),
    "spmv_axpby": (C.c_int, [_vp, C.c_double, _vp, C.c_double, _vp, _vp]),
    "spmv_apply_dot": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _f64p]),
    "spmv_cg": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _f64p]),
    "spmv_symgs": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32]),
    "spmv_symgs_setup": (C.c_int, [_vp, _vp]),
    "spmv_symgs_order": (C.c_int, [_vp, _vp, _i32p]),
    "spmv_coo_to_csr": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "spmv_coo_to_ell": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "spmv_csr_to_ell": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "spmv_csr_split_columns": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, C.POINTER(_vp), C.POINTER(_vp)]),
    "spmv_partition_rows": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, _i64p, _i64p]),
    "spmv_partition_rows_balanced": (C.c_int, [C.c_int64, _vp, C.c_int32, _vp]),
    "spmv_gen_csr_uniform": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "spmv_gen_ell_banded": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "spmv_gen_dia_banded": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "spmv_gen_coo_powerlaw": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "spmv_gen_coo_powerlaw_sorted": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(_vp)]),
    "spmv_mat_partition_rows": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "spmv_ctx_xcd_round_robin": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "spmv_apply_host": (C.c_int, [_vp, _vp, _vp, _vp]),
    "spmv_csr_extract_rows": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "spmv_gen_vec_uniform": (C.c_int, [_vp, _vp, C.c_int64, C.c_uint64]),
}


class SpmvError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libspmv_hip status {code}: {message}")
        self.code = code


_lib = None


def load(path: os.PathLike | None = None) -> C.CDLL:
    """Load libspmv_hip.so (once) and declare every prototype.  Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise FileNotFoundError(
            f"{p} not found: build the HIP engine first (make engine, or __graft_entry__.build()). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != 0:
        raise SpmvError(rc, load().spmv_last_error().decode("utf-8", "replace"))


def _host(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a) -> int | None:
    """address of a numpy array / torch tensor / raw int; None passes NULL"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return int(a)


def device_count() -> int:
    n = C.c_int(0)
    rc = load().spmv_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def plan_check(plan: bytes) -> tuple[int, int, int]:
    """(root format, root kernel, number of nodes) of a plan blob, or SpmvError with the reason; needs no device (spmv_plan_check)"""
    f, k, n = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _check(load().spmv_plan_check(plan, len(plan), C.byref(f), C.byref(k), C.byref(n)))
    return f.value, k.value, n.value


def partition_rows(nrow: int, nparts: int, part: int) -> tuple[int, int]:
    """Equal rows per part, last part takes the remainder (reference src/mat_vec.cpp:233,245-246)."""
    b, e = C.c_int64(), C.c_int64()
    _check(load().spmv_partition_rows(nrow, nparts, part, C.byref(b), C.byref(e)))
    return b.value, e.value


def partition_rows_balanced(row_ptr64, nparts: int) -> np.ndarray:
    rp = _host(row_ptr64, np.int64)
    bounds = np.zeros(nparts + 1, dtype=np.int64)
    _check(load().spmv_partition_rows_balanced(len(rp) - 1, rp.ctypes.data, nparts, bounds.ctypes.data))
    return bounds


class Context:
    """One HIP device + one stream (spmv_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = load()
        h = _vp()
        if stream is None:
            _check(self._lib.spmv_ctx_create(device, C.byref(h)))
        else:
            _check(self._lib.spmv_ctx_create_on_stream(device, _vp(stream), C.byref(h)))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._lib.spmv_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(self._lib.spmv_sync(self.h))

    def apply_host(self, A: "Matrix", x_host: np.ndarray, y_host: np.ndarray) -> None:
        """y_host += A * x_host with HOST arrays (float64, contiguous), synchronous: the reference's call shape in one entry point"""
        if x_host.dtype != np.float64 or y_host.dtype != np.float64 or not x_host.flags.c_contiguous or not y_host.flags.c_contiguous:
            raise ValueError("apply_host: float64 contiguous arrays")
        i = A.info
        if x_host.size != i.ncol or y_host.size != i.nrow:
            raise ValueError(f"apply_host: x has {x_host.size} entries (ncol {i.ncol}), y {y_host.size} (nrow {i.nrow})")
        _check(self._lib.spmv_apply_host(self.h, A.h, x_host.ctypes.data, y_host.ctypes.data))

    def get_param(self, name: str) -> int:
        """what the context found out about its platform (spmv_ctx_get_param): "host_stores", "xcd_round_robin", ..."""
        v = C.c_int64(0)
        _check(self._lib.spmv_ctx_get_param(self.h, name.encode(), C.byref(v)))
        return v.value

    def set_plan(self, plan: bytes | None) -> None:
        """handles of the plan's format created on this context from now on take `plan` instead of selecting; None clears it"""
        if plan is None:
            _check(self._lib.spmv_ctx_set_plan(self.h, None, 0))
        else:
            _check(self._lib.spmv_ctx_set_plan(self.h, plan, len(plan)))

    def xcd_round_robin(self) -> tuple[int, int]:
        """(1 / 0 / -1, distinct XCD ids seen): the start-up probe of workgroup placement (spmv_ctx_xcd_round_robin)"""
        rr, seen = C.c_int32(0), C.c_int32(0)
        _check(self._lib.spmv_ctx_xcd_round_robin(self.h, C.byref(rr), C.byref(seen)))
        return rr.value, seen.value

    # ---- vectors
    def vector(self, n: int) -> "Vector":
        h = _vp()
        _check(self._lib.spmv_vec_create(self.h, n, C.byref(h)))
        return Vector(self, h, n)

    def vector_from(self, host) -> "Vector":
        a = _host(host, np.float64)
        v = self.vector(a.size)
        v.upload(a)
        return v

    def wrap_vector(self, dev, n: int | None = None) -> "Vector":
        """borrow device memory (torch tensor or raw pointer)"""
        if n is None:
            n = dev.numel()
        h = _vp()
        _check(self._lib.spmv_vec_wrap_device(self.h, n, _vp(_ptr(dev)), C.byref(h)))
        v = Vector(self, h, n)
        v._keep = dev
        return v

    # ---- matrices from host arrays
    def csr(self, nrow, ncol, row_ptr, col, val) -> "Matrix":
        rp, c, v = _host(row_ptr, np.int32), _host(col, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_csr_upload(self.h, nrow, ncol, _ptr(rp), _ptr(c), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    def csr_shard(self, row_begin, row_end, ncol, row_ptr64, col, val) -> "Matrix":
        rp, c, v = _host(row_ptr64, np.int64), _host(col, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_csr_upload_shard(self.h, row_begin, row_end, ncol, _ptr(rp), _ptr(c), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    def coo(self, nrow, ncol, row, col, val) -> "Matrix":
        r, c, v = _host(row, np.int32), _host(col, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_coo_upload(self.h, nrow, ncol, r.size, _ptr(r), _ptr(c), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    def ell(self, nrow, ncol, k, nnz, col, val) -> "Matrix":
        c, v = _host(col, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_ell_upload(self.h, nrow, ncol, k, nnz, _ptr(c), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    def csc(self, nrow, ncol, col_ptr, row, val) -> "Matrix":
        cp, r, v = _host(col_ptr, np.int32), _host(row, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_csc_upload(self.h, nrow, ncol, _ptr(cp), _ptr(r), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    def dia(self, nrow, ncol, offsets, val) -> "Matrix":
        o, v = _host(offsets, np.int32), _host(val, np.float64)
        h = _vp()
        _check(self._lib.spmv_dia_upload(self.h, nrow, ncol, o.size, _ptr(o), _ptr(v), C.byref(h)))
        return Matrix(self, h)

    # ---- matrices borrowing device memory
    def wrap_csr(self, nrow, ncol, d_row_ptr, d_col, d_val) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_csr_wrap_device(self.h, nrow, ncol, _ptr(d_row_ptr), _ptr(d_col), _ptr(d_val), C.byref(h)))
        m = Matrix(self, h)
        m._keep = (d_row_ptr, d_col, d_val)
        return m

    def wrap_coo(self, nrow, ncol, nnz, d_row, d_col, d_val) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_coo_wrap_device(self.h, nrow, ncol, nnz, _ptr(d_row), _ptr(d_col), _ptr(d_val), C.byref(h)))
        m = Matrix(self, h)
        m._keep = (d_row, d_col, d_val)
        return m

    def wrap_ell(self, nrow, ncol, k, nnz, d_col, d_val) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_ell_wrap_device(self.h, nrow, ncol, k, nnz, _ptr(d_col), _ptr(d_val), C.byref(h)))
        m = Matrix(self, h)
        m._keep = (d_col, d_val)
        return m

    # ---- generators
    def gen_csr_uniform(self, row_begin, row_end, ncol, k, band=0, seed=1) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_gen_csr_uniform(self.h, row_begin, row_end, ncol, k, band, seed, C.byref(h)))
        return Matrix(self, h)

    def gen_ell_banded(self, nrow, ncol, k, seed=1) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_gen_ell_banded(self.h, nrow, ncol, k, seed, C.byref(h)))
        return Matrix(self, h)

    def gen_dia_banded(self, nrow, k, seed=1) -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_gen_dia_banded(self.h, nrow, k, seed, C.byref(h)))
        return Matrix(self, h)

    def gen_coo_powerlaw(self, nrow, ncol, max_len=4096, seed=1, sorted_by_length=False) -> "Matrix":
        """row-sorted COO with power-law row lengths; sorted_by_length: the lengths at the distribution's quantiles, longest first"""
        h = _vp()
        fn = self._lib.spmv_gen_coo_powerlaw_sorted if sorted_by_length else self._lib.spmv_gen_coo_powerlaw
        _check(fn(self.h, nrow, ncol, max_len, seed, C.byref(h)))
        return Matrix(self, h)

    def extract_rows(self, csr: "Matrix", row_begin: int, row_end: int) -> "Matrix":
        """rows [row_begin, row_end) of a device-resident CSR handle (of any context) as a shard on THIS context"""
        h = _vp()
        _check(self._lib.spmv_csr_extract_rows(self.h, csr.h, row_begin, row_end, C.byref(h)))
        return Matrix(self, h)

    def gen_vector(self, n, index_offset=0, seed=1) -> "Vector":
        v = self.vector(n)
        _check(self._lib.spmv_gen_vec_uniform(self.h, v.h, index_offset, seed))
        return v

    # ---- ops
    def apply(self, A: "Matrix", x: "Vector", y: "Vector") -> None:
        """y += A*x, asynchronous on the context's stream"""
        _check(self._lib.spmv_apply(self.h, A.h, x.h, y.h))

    def apply_timed(self, A: "Matrix", x: "Vector", y: "Vector", reps: int) -> float:
        ms = C.c_double(0.0)
        _check(self._lib.spmv_apply_timed(self.h, A.h, x.h, y.h, reps, C.byref(ms)))
        return ms.value

    def dot(self, x: "Vector", y: "Vector") -> float:
        r = C.c_double(0.0)
        _check(self._lib.spmv_dot(self.h, x.h, y.h, C.byref(r)))
        return r.value

    def axpby(self, alpha: float, x: "Vector", beta: float, y: "Vector", w: "Vector") -> None:
        _check(self._lib.spmv_axpby(self.h, alpha, x.h, beta, y.h, w.h))

    def mem_info(self) -> tuple[int, int]:
        """(free, total) device memory in bytes"""
        f, t = C.c_int64(0), C.c_int64(0)
        _check(self._lib.spmv_ctx_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def apply_dot(self, A: "Matrix", x: "Vector", y: "Vector", w: "Vector", overwrite: bool = False) -> float:
        """y = A*x (overwrite) or y += A*x; returns w . y of the updated y (fused into the product where possible)"""
        d = C.c_double(0.0)
        _check(self._lib.spmv_apply_dot(self.h, A.h, x.h, y.h, 1 if overwrite else 0, w.h, C.byref(d)))
        return d.value

    def symgs(self, A: "Matrix", b: "Vector", x: "Vector", sweeps: int = 1) -> None:
        """`sweeps` symmetric Gauss-Seidel sweeps on A x = b, x updated in place (CSR handle holding the whole square
        matrix; the first call analyses it).  Sweep order: A.set_param("symgs_order", 1 multicolour (default) | 0 rows)"""
        _check(self._lib.spmv_symgs(self.h, A.h, b.h, x.h, sweeps))

    def symgs_order(self, A: "Matrix"):
        """sets the handle up if need be and returns order[k] = the k-th row of a forward sweep"""
        _check(self._lib.spmv_symgs_setup(self.h, A.h))
        out = np.zeros(A.info.nrow, dtype=np.int32)
        _check(self._lib.spmv_symgs_order(self.h, A.h, out.ctypes.data_as(_i32p)))
        return out

    def cg(self, A: "Matrix", b: "Vector", x: "Vector", max_iter: int = 1000, rel_tol: float = 1e-8, check_every: int = 1,
           jacobi: bool = False, symgs: bool = False):
        """conjugate gradients on the device from the x passed in (jacobi: diagonal preconditioner, symgs: one symmetric
        Gauss-Seidel sweep per iteration; CSR handles); returns (iterations, ||r|| / ||b||)"""
        it, res = C.c_int32(0), C.c_double(0.0)
        _check(self._lib.spmv_cg(self.h, A.h, b.h, x.h, max_iter, rel_tol, check_every, PRECOND_SYMGS if symgs else (PRECOND_JACOBI if jacobi else PRECOND_NONE), C.byref(it),
                                 C.byref(res)))
        return it.value, res.value

    def coo_to_csr(self, coo: "Matrix") -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_coo_to_csr(self.h, coo.h, C.byref(h)))
        return Matrix(self, h)

    def csr_split_columns(self, csr: "Matrix", col_begin: int, col_end: int):
        """(inside, outside): entries with a column in [col_begin, col_end) rebased to 0, and the rest with global columns"""
        a, b = _vp(), _vp()
        _check(self._lib.spmv_csr_split_columns(self.h, csr.h, col_begin, col_end, C.byref(a), C.byref(b)))
        return Matrix(self, a), Matrix(self, b)

    def coo_to_ell(self, coo: "Matrix") -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_coo_to_ell(self.h, coo.h, C.byref(h)))
        return Matrix(self, h)

    def csr_to_ell(self, csr: "Matrix") -> "Matrix":
        h = _vp()
        _check(self._lib.spmv_csr_to_ell(self.h, csr.h, C.byref(h)))
        return Matrix(self, h)


class Comm:
    """exchange between several contexts of this process (spmv_comm_*): the x all-gather of the sharded drivers"""

    def __init__(self, ctxs):
        self.ctxs = list(ctxs)
        self._lib = self.ctxs[0]._lib
        arr = (_vp * len(self.ctxs))(*[c.h for c in self.ctxs])
        self.h = _vp()
        _check(self._lib.spmv_comm_create(arr, len(self.ctxs), C.byref(self.h)))

    def __del__(self):
        try:
            if self.h:
                self._lib.spmv_comm_destroy(self.h)
        except Exception:
            pass
        self.h = None

    @property
    def backend(self) -> str:
        return self._lib.spmv_comm_backend(self.h).decode()

    def allgather(self, vecs, offsets) -> None:
        n = len(self.ctxs)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        # the C side reads vecs[0..n) and offsets[0..n]: a short array would be read out of bounds on the host
        if len(vecs) != n or off.ndim != 1 or off.size != n + 1:
            raise ValueError(f"Comm.allgather: {n} participants need {n} vectors and {n + 1} offsets, got {len(vecs)} and {off.size}")
        arr = (_vp * n)(*[v.h for v in vecs])
        _check(self._lib.spmv_comm_allgather(self.h, arr, off.ctypes.data_as(_i64p)))


class Vector:
    def __init__(self, ctx: Context, h, n: int):
        self.ctx, self.h, self.n = ctx, h, n
        self._keep = None

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx._lib.spmv_vec_destroy(self.h)
        except Exception:
            pass
        self.h = None

    def upload(self, host, offset: int = 0) -> None:
        a = _host(host, np.float64)
        _check(self.ctx._lib.spmv_vec_upload(self.h, offset, a.size, _ptr(a)))

    def download(self, offset: int = 0, n: int | None = None) -> np.ndarray:
        n = self.n - offset if n is None else n
        out = np.empty(n, dtype=np.float64)
        _check(self.ctx._lib.spmv_vec_download(self.h, offset, n, _ptr(out)))
        return out

    def fill(self, a: float) -> None:
        _check(self.ctx._lib.spmv_vec_fill(self.h, a))

    def copy_from(self, src: "Vector", n: int, dst_offset: int = 0, src_offset: int = 0) -> None:
        _check(self.ctx._lib.spmv_vec_copy(self.h, dst_offset, src.h, src_offset, n))

    @property
    def device_ptr(self) -> int:
        p = _vp()
        _check(self.ctx._lib.spmv_vec_device_ptr(self.h, C.byref(p)))
        return p.value or 0


class Matrix:
    def __init__(self, ctx: Context, h):
        self.ctx, self.h = ctx, h
        self._keep = None

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx._lib.spmv_mat_destroy(self.h)
        except Exception:
            pass
        self.h = None

    def validate(self) -> None:
        """raise SpmvError if an index is out of range or an offset array is inconsistent"""
        _check(self.ctx._lib.spmv_mat_validate(self.h))

    @property
    def info(self) -> MatInfo:
        i = MatInfo()
        _check(self.ctx._lib.spmv_mat_get_info(self.h, C.byref(i)))
        return i

    def set_kernel(self, kernel: int, lanes_per_row: int = 0) -> None:
        _check(self.ctx._lib.spmv_mat_set_kernel(self.h, kernel, lanes_per_row))

    def set_param(self, name: str, value: int) -> None:
        _check(self.ctx._lib.spmv_mat_set_param(self.h, name.encode(), value))

    def get_param(self, name: str) -> int:
        v = C.c_int64(0)
        _check(self.ctx._lib.spmv_mat_get_param(self.h, name.encode(), C.byref(v)))
        return v.value

    def get_plan(self) -> bytes:
        """the handle's set-up decisions (kernel, layout, tuned parameters; its copies' too) as a POD blob (spmv_mat_get_plan)"""
        n = C.c_int64(0)
        _check(self.ctx._lib.spmv_mat_get_plan(self.h, None, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        _check(self.ctx._lib.spmv_mat_get_plan(self.h, buf, C.byref(n)))
        return buf.raw[: n.value]

    def set_plan(self, plan: bytes) -> None:
        """build exactly the kernel and layout of `plan` (from get_plan of a handle of the same format), no timing launch"""
        _check(self.ctx._lib.spmv_mat_set_plan(self.h, plan, len(plan)))

    def set_flags(self, flags: int) -> None:
        _check(self.ctx._lib.spmv_mat_set_flags(self.h, flags))

    def partition_rows(self, nparts: int, balance_entries: bool = False) -> np.ndarray:
        """bounds[nparts + 1] of a row partition of this handle (columns for CSC): equal rows (the reference's split) or by entries"""
        bounds = np.zeros(nparts + 1, dtype=np.int64)
        _check(self.ctx._lib.spmv_mat_partition_rows(self.h, nparts, 1 if balance_entries else 0, bounds.ctypes.data))
        return bounds

    def download(self):
        """(a, b, v) host copies; see spmv_mat_download for which array is which per format"""
        i = self.info
        fmt = i.format
        if fmt == FMT_CSR:
            na, nb, nv = i.nrow + 1, i.nnz, i.nnz
        elif fmt == FMT_COO:
            na = nb = nv = i.nnz
        elif fmt == FMT_ELL:
            na, nb, nv = 0, i.nrow * i.ell_k, i.nrow * i.ell_k
        elif fmt == FMT_CSC:
            na, nb, nv = i.ncol + 1, i.nnz, i.nnz
        else:
            na, nb, nv = i.ell_k, 0, i.nrow * i.ell_k
        a = np.empty(na, dtype=np.int32)
        b = np.empty(nb, dtype=np.int32)
        v = np.empty(nv, dtype=np.float64)
        _check(self.ctx._lib.spmv_mat_download(self.h, _ptr(a) if na else None, _ptr(b) if nb else None, _ptr(v) if nv else None))
        return a, b, v
