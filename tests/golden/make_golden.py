"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libarmspmv_ref.so).

Run in the build container only (the reference sources live at /root/reference there):
    make ref && python tests/golden/make_golden.py
The fixtures are DATA: inputs (or, for c1, the seed + a sha256 of the inputs) and the outputs the
reference's own compiled code produced for them — y after 1 and after 50 accumulating calls
(NUM_TEST, main.cpp:16) of COOMatirxMatVector / CSRMatrixMatVector / ELLMatrixMatVector /
CSCMatrixMatVector (+ DIAMatrixMatVector on the square tridiagonal), and the arrays built by the
converting constructors CSRMatrix(COO), CSCMatrix(COO), ELLMatrix(COO), DIAMatrix(CSR).
OMP_NUM_THREADS=1 so that the `omp atomic` loops (COO, CSC) run in file order.
"""
import os

os.environ["OMP_NUM_THREADS"] = "1"

import ctypes as C
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))

import cases  # noqa: E402
import oracle_lib as ol  # noqa: E402

_p = ol._p
NUM_TEST = 50


def run_case(ref, c):
    nrow, ncol = c["nrow"], c["ncol"]
    row, col, val, x = ol.i32(c["row"]), ol.i32(c["col"]), ol.f64(c["val"]), ol.f64(c["x"])
    nnz = len(val)
    out = {}

    # converting constructors
    rp = np.zeros(nrow + 1, np.int32)
    cc = np.zeros(nnz, np.int32)
    cv = np.zeros(nnz, np.float64)
    ref.ref_coo_to_csr(nrow, ncol, nnz, _p(row), _p(col), _p(val), _p(rp), _p(cc), _p(cv))
    cp = np.zeros(ncol + 1, np.int32)
    cr = np.zeros(nnz, np.int32)
    cw = np.zeros(nnz, np.float64)
    ref.ref_coo_to_csc(nrow, ncol, nnz, _p(row), _p(col), _p(val), _p(cp), _p(cr), _p(cw))
    k = ref.ref_coo_to_ell(nrow, ncol, nnz, _p(row), _p(col), _p(val), None, None)
    ec = np.zeros(nrow * k, np.int32)
    ev = np.zeros(nrow * k, np.float64)
    ref.ref_coo_to_ell(nrow, ncol, nnz, _p(row), _p(col), _p(val), _p(ec), _p(ev))

    def accumulate(fn):
        y = np.zeros(nrow, np.float64)
        fn(y)
        y1 = y.copy()
        for _ in range(NUM_TEST - 1):
            fn(y)
        return y1, y

    out["y1_coo"], out["y50_coo"] = accumulate(lambda y: ref.ref_coo_spmv(nrow, ncol, nnz, _p(row), _p(col), _p(val), _p(x), _p(y)))
    out["y1_csr"], out["y50_csr"] = accumulate(lambda y: ref.ref_csr_spmv(nrow, ncol, _p(rp), _p(cc), _p(cv), _p(x), _p(y)))
    out["y1_ell"], out["y50_ell"] = accumulate(lambda y: ref.ref_ell_spmv(nrow, ncol, nnz, k, _p(ec), _p(ev), _p(x), _p(y)))
    out["y1_csc"], out["y50_csc"] = accumulate(lambda y: ref.ref_csc_spmv(nrow, ncol, _p(cp), _p(cr), _p(cw), _p(x), _p(y)))

    if c["name"] == "c1":
        # inputs are regenerated from the seed; keep only digests of them and of the converted arrays
        out["seed"] = np.int64(cases.C1_SEED)
        out["sha_inputs"] = np.array(cases.digest(row, col, val, x))
        out["sha_csr"] = np.array(cases.digest(rp, cc, cv))
        out["sha_csc"] = np.array(cases.digest(cp, cr, cw))
        out["sha_ell"] = np.array(cases.digest(ec, ev))
        out["ell_k"] = np.int32(k)
    else:
        out.update(nrow=np.int32(nrow), ncol=np.int32(ncol), row=row, col=col, val=val, x=x)
        out.update(csr_row_ptr=rp, csr_col=cc, csr_val=cv, csc_col_ptr=cp, csc_row=cr, csc_val=cw,
                   ell_k=np.int32(k), ell_col=ec, ell_val=ev)
    if c["name"] == "tri8":
        nd = ref.ref_csr_to_dia(nrow, ncol, _p(rp), _p(cc), _p(cv), None, None)
        off = np.zeros(nd, np.int32)
        dv = np.zeros(nrow * nd, np.float64)
        ref.ref_csr_to_dia(nrow, ncol, _p(rp), _p(cc), _p(cv), _p(off), _p(dv))
        out["dia_offsets"], out["dia_val"] = off, dv
        out["y1_dia"], out["y50_dia"] = accumulate(lambda y: ref.ref_dia_spmv(nrow, ncol, nd, _p(off), _p(dv), _p(x), _p(y)))
        # BLAS-1 (src/vec_vec.cpp) on the same vectors
        ref.ref_dot.restype = C.c_double
        out["dot_xx"] = np.float64(ref.ref_dot(nrow, _p(x), _p(x)))
        yv = out["y1_csr"].copy()
        for tag, (a, b) in dict(g=(0.75, -1.25), a0=(0.0, 2.0), b0=(3.0, 0.0), a1=(1.0, 0.5), am1=(-1.0, 0.5),
                                b1=(0.5, 1.0), bm1=(0.5, -1.0)).items():
            w = np.zeros(nrow)
            ref.ref_axpby(nrow, C.c_double(a), _p(x), C.c_double(b), _p(yv), _p(w))
            out[f"axpby_{tag}"] = w
    return out


def main():
    if not ol.ref_available():
        sys.exit("oracle/_ref/libarmspmv_ref.so missing: run `make ref` in the build container first")
    ref = ol.load_ref()
    for make in cases.ALL_CASES:
        c = make()
        out = run_case(ref, c)
        path = HERE / f"{c['name']}.npz"
        np.savez_compressed(path, **out)
        print(f"wrote {path.name}: {sorted(out)[:6]}... ({path.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
