"""Host logic of the drop-in layer that needs no GPU: the Matrix Market reader (reference src/data_io.cpp:45-105
behaviour: comments skipped, 1-based -> 0-based, file order kept, size line = rows cols entries), the vector text
files and the timer.  Called through the C++ symbols of libarmspmv_compat.so (the reference exports C++ names too)."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases

ROOT = Path(__file__).resolve().parent.parent
LIB = Path(os.environ.get("SPMV_COMPAT_SO") or ROOT / "arm-spmv_amd" / "lib" / "libarmspmv_compat.so")  # override: sanitizer build


class COO(C.Structure):  # include/arm_spmv_compat.hpp == reference include/matrix.h:7-16
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_int), ("row_ind", C.POINTER(C.c_int)),
                ("col_ind", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_double))]


class Vec(C.Structure):  # reference include/vector.h:7-8
    _fields_ = [("size", C.c_int), ("values", C.POINTER(C.c_double))]


@pytest.fixture(scope="module")
def lib():
    if not LIB.exists():
        subprocess.run(["make", "host"], cwd=ROOT, check=True, capture_output=True)
    return C.CDLL(str(LIB))


def _write_mtx(path, c, comment=True):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        if comment:
            f.write("% a comment line\n%another\n")
        f.write(f"{c['nrow']} {c['ncol']} {len(c['val'])}\n")
        for r, cc, v in zip(c["row"], c["col"], c["val"]):
            f.write(f"{r + 1} {cc + 1} {v:.17g}\n")


@pytest.mark.parametrize("threads", ["1", "5"])
@pytest.mark.parametrize("make", cases.ALL_CASES, ids=lambda f: f.__name__)
def test_matrix_market_round_trip(lib, tmp_path, make, threads, monkeypatch):
    """threads=1: the reference's fscanf loop; threads=5: the parallel token parser (files >= 65536 entries: c1)"""
    monkeypatch.setenv("SPMV_MTX_THREADS", threads)
    c = make()
    p = tmp_path / "m.mtx"
    _write_mtx(p, c)
    read = getattr(lib, "_Z13COOMatrixReadPKcR9COOMatrix")
    read.argtypes = [C.c_char_p, C.POINTER(COO)]
    A = COO()
    read(str(p).encode(), C.byref(A))
    assert (A.nrow, A.ncol, A.nnz) == (c["nrow"], c["ncol"], len(c["val"]))
    n = A.nnz
    assert np.array_equal(np.ctypeslib.as_array(A.row_ind, (n,)), c["row"])  # 0-based, file order
    assert np.array_equal(np.ctypeslib.as_array(A.col_ind, (n,)), c["col"])
    assert np.array_equal(np.ctypeslib.as_array(A.values, (n,)), c["val"])  # %.17g is round-trip exact


def test_vector_files_and_timer(lib, tmp_path):
    write = getattr(lib, "_Z11VectorWritePKcRK6Vector")
    read = getattr(lib, "_Z10VectorReadPKcR6Vector")
    write.argtypes = [C.c_char_p, C.POINTER(Vec)]
    read.argtypes = [C.c_char_p, C.POINTER(Vec)]
    x = np.array([0.5, -1.25, 3.0, 1e-3, 0.1])
    v = Vec(len(x), x.ctypes.data_as(C.POINTER(C.c_double)))
    p = tmp_path / "v.txt"
    write(str(p).encode(), C.byref(v))
    assert p.read_text().split("\n")[0] == "5"
    back = Vec()
    read(str(p).encode(), C.byref(back))
    got = np.ctypeslib.as_array(back.values, (back.size,))
    assert back.size == 5 and np.allclose(got, x, rtol=1e-15)  # "%20.16g": 16 significant digits (src/data_io.cpp:37)
    timer = getattr(lib, "_Z7mytimerv")
    timer.restype = C.c_double
    assert timer() == 0.0  # the first call returns 0.0 (src/mytime.cpp:10-15)
    assert timer() >= 0.0


def test_reader_rejects_bad_input_like_the_reference(tmp_path):
    """missing file / bad banner / complex matrices -> message + exit(1) (src/data_io.cpp:53-71); run in a child"""
    bad = tmp_path / "bad.mtx"
    bad.write_text("not a banner\n1 1 1\n1 1 1.0\n")
    cplx = tmp_path / "c.mtx"
    cplx.write_text("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n")
    for path, msg in ((tmp_path / "none.mtx", "Failed to open"), (bad, "Could not process Matrix Market banner"),
                      (cplx, "does not support")):
        code = ("import ctypes as C; lib=C.CDLL(%r); f=getattr(lib,'_Z13COOMatrixReadPKcR9COOMatrix');"
                "buf=(C.c_char*64)(); f(%r, buf)") % (str(LIB), str(path).encode())
        r = subprocess.run(["python3", "-c", code], capture_output=True, text=True)
        assert r.returncode == 1 and msg in r.stdout, (path, r.stdout, r.stderr)


@pytest.mark.parametrize("threads", ["1", "4"])
def test_opt_in_pattern_and_symmetric_expansion(lib, tmp_path, threads, monkeypatch):
    """by default a symmetric file is NOT expanded (as in the reference, src/data_io.cpp:83-88);
    SPMV_MTX_SYMMETRIC=1 adds the mirrored entries, SPMV_MTX_PATTERN=1 reads `pattern` files as (i, j) pairs of 1.0"""
    monkeypatch.setenv("SPMV_MTX_THREADS", threads)
    rng = np.random.default_rng(3)
    n, nz = 700, 70_000  # >= 65536 entries: the parallel parser takes the file when threads > 1
    i = rng.integers(0, n, nz)
    j = rng.integers(0, n, nz)
    i, j = np.maximum(i, j), np.minimum(i, j)  # lower triangle, diagonal included
    v = rng.uniform(-1, 1, nz)
    read = getattr(lib, "_Z13COOMatrixReadPKcR9COOMatrix")
    read.argtypes = [C.c_char_p, C.POINTER(COO)]

    def load(path):
        A = COO()
        read(str(path).encode(), C.byref(A))
        k = A.nnz
        return (A.nrow, A.ncol, np.ctypeslib.as_array(A.row_ind, (k,)).copy(), np.ctypeslib.as_array(A.col_ind, (k,)).copy(),
                np.ctypeslib.as_array(A.values, (k,)).copy())

    for sym, sign in (("symmetric", 1.0), ("skew-symmetric", -1.0)):
        p = tmp_path / f"{sym}.mtx"
        with open(p, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate real {sym}\n{n} {n} {nz}\n")
            for a, b, c in zip(i, j, v):
                f.write(f"{a + 1} {b + 1} {c:.17g}\n")
        monkeypatch.delenv("SPMV_MTX_SYMMETRIC", raising=False)
        _, _, r, c, w = load(p)
        assert len(w) == nz and np.array_equal(r, i) and np.array_equal(c, j)  # default: as the reference
        monkeypatch.setenv("SPMV_MTX_SYMMETRIC", "1")
        _, _, r, c, w = load(p)
        off = i != j
        assert len(w) == nz + off.sum()
        # every stored entry keeps its place in file order, followed by its mirror
        er = np.empty(len(w), np.int64)
        ec = np.empty(len(w), np.int64)
        ev = np.empty(len(w))
        pos = np.arange(nz) + np.concatenate(([0], np.cumsum(off)[:-1]))
        er[pos], ec[pos], ev[pos] = i, j, v
        er[pos[off] + 1], ec[pos[off] + 1], ev[pos[off] + 1] = j[off], i[off], sign * v[off]
        assert np.array_equal(r, er) and np.array_equal(c, ec) and np.array_equal(w, ev)
    monkeypatch.delenv("SPMV_MTX_SYMMETRIC", raising=False)
    p = tmp_path / "pattern.mtx"
    with open(p, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate pattern general\n{n} {n} {nz}\n")
        for a, b in zip(i, j):
            f.write(f"{a + 1} {b + 1}\n")
    monkeypatch.setenv("SPMV_MTX_PATTERN", "1")
    _, _, r, c, w = load(p)
    assert len(w) == nz and np.array_equal(r, i) and np.array_equal(c, j) and np.all(w == 1.0)


def test_binary_cache_round_trip(lib, tmp_path, monkeypatch):
    """SPMV_MTX_CACHE=1: the second read comes from `<file>.spmvbin` and equals the first; a changed .mtx or changed
    options invalidate it"""
    import time

    c = cases.case_c1() if hasattr(cases, "case_c1") else cases.ALL_CASES[-1]()
    p = tmp_path / "m.mtx"
    _write_mtx(p, c)
    read = getattr(lib, "_Z13COOMatrixReadPKcR9COOMatrix")
    read.argtypes = [C.c_char_p, C.POINTER(COO)]

    def load():
        A = COO()
        read(str(p).encode(), C.byref(A))
        n = A.nnz
        return (A.nrow, A.ncol, np.ctypeslib.as_array(A.row_ind, (n,)).copy(), np.ctypeslib.as_array(A.col_ind, (n,)).copy(),
                np.ctypeslib.as_array(A.values, (n,)).copy())

    monkeypatch.delenv("SPMV_MTX_CACHE", raising=False)
    first = load()
    assert not (tmp_path / "m.mtx.spmvbin").exists()
    monkeypatch.setenv("SPMV_MTX_CACHE", "1")
    second = load()
    cache = tmp_path / "m.mtx.spmvbin"
    assert cache.exists()
    stamp = cache.stat().st_mtime_ns
    third = load()  # from the cache
    assert cache.stat().st_mtime_ns == stamp
    for a, b in ((first, second), (first, third)):
        assert a[0] == b[0] and a[1] == b[1] and all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))
    # the source changes (one more entry): the cache no longer matches and is rewritten
    time.sleep(1.1)
    c2 = dict(c, row=np.append(c["row"], 0), col=np.append(c["col"], 0), val=np.append(c["val"], 2.5))
    _write_mtx(p, c2)
    fourth = load()
    assert len(fourth[4]) == len(first[4]) + 1 and fourth[4][-1] == 2.5
