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


# ---------------------------------------------------------------------------------- differential: reader vs the reference's
REF_SO = ROOT / "oracle" / "_ref" / "libarmspmv_ref.so"


def _odd_files(tmp_path):
    """Matrix Market files whose reading is DEFINED in the reference (every `fscanf("%d %d %lg\\n")` of
    src/data_io.cpp:83-88 converts three tokens) but that a reader written from the format description would treat
    differently.  (Files with fewer than 3 x nnz tokens are left out: there the reference returns uninitialised heap
    for the missing entries, the shim stops with a message — no common answer exists.)"""
    rng = np.random.RandomState(11)
    files = {}

    def put(name, text):
        p = tmp_path / name
        p.write_text(text)
        files[name] = p

    put("comments_blank_tabs.mtx",
        "%%MatrixMarket matrix coordinate real general\n% c1\n%\n%%% c3\n\n  5 4   6  \n\n1\t1\t 1.5\n\n2 3 -2.25   \n\n\n"
        "5 4 1e-3\n3 2 +4\n 4 1 .5\n1 4 -7.0E+2\n% a trailing comment is not an entry\n9 9 9\n")
    put("integer_field.mtx", "%%MatrixMarket matrix coordinate integer general\n3 3 4\n1 1 7\n2 2 -3\n3 1 12\n1 3 0\n")
    # symmetric / skew / hermitian-free variants: read as stored, NOT expanded
    put("symmetric.mtx", "%%MatrixMarket matrix coordinate real symmetric\n4 4 5\n1 1 2.0\n2 1 -1.0\n3 2 -1.0\n4 4 3.0\n4 1 0.25\n")
    put("skew.mtx", "%%MatrixMarket matrix coordinate real skew-symmetric\n3 3 2\n2 1 5.0\n3 1 -6.0\n")
    # four integer tokens per line: the reference re-chunks the token stream in threes, whatever the lines look like
    lines = "".join(f"{i + 1} {(i * 7) % 9 + 1} {i - 4} {i + 100}\n" for i in range(9))
    put("extra_column.mtx", "%%MatrixMarket matrix coordinate real general\n9 9 9\n" + lines)
    # a `pattern` file that declares fewer entries than it has tokens for: (i, j, next i) triples, all conversions succeed
    put("pattern_short_count.mtx", "%%MatrixMarket matrix coordinate pattern general\n6 6 4\n1 2\n3 4\n5 6\n6 1\n2 2\n4 4\n")
    # what %lg accepts: hex floats, infinities, nan, many digits
    put("float_forms.mtx", "%%MatrixMarket matrix coordinate real general\n2 8 8\n1 1 0x1.8p1\n1 2 inf\n1 3 -INF\n1 4 nan\n"
        "1 5 123456789012345678901234567890\n1 6 1e-400\n1 7 0.1000000000000000055511151231257827\n1 8 -0\n")
    put("more_lines_than_declared.mtx", "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1\n2 2 2\n3 3 3\n1 2 4\n")
    put("zero_entries.mtx", "%%MatrixMarket matrix coordinate real general\n7 5 0\n")
    put("array_of_ints_as_indices.mtx", "%%MatrixMarket matrix coordinate real general\n10 10 3\n007 +3 1\n10 010 2\n1 1 3\n")
    # big enough for the parallel parser (>= 65536 entries), with ragged formatting
    n = 70_001
    r, c = rng.randint(1, 5000, n), rng.randint(1, 4000, n)
    v = rng.uniform(-1, 1, n)
    seps = [" ", "\t", "  ", " \t "]
    body = []
    for k in range(n):
        s = seps[k % 4]
        val = ("%.17g" % v[k]) if k % 5 else ("%+.6e" % v[k])
        body.append(f"{r[k]}{s}{c[k]}{s}{val}" + ("\n\n" if k % 97 == 0 else "\n"))
    put("large_ragged.mtx", f"%%MatrixMarket matrix coordinate real general\n% big\n5000 4000 {n}\n" + "".join(body))
    big_sym = "".join(f"{max(a, b)} {min(a, b)} {k % 13 - 6}\n" for k, (a, b) in enumerate(zip(r, c)))
    put("large_symmetric_integer.mtx", f"%%MatrixMarket matrix coordinate integer symmetric\n5000 5000 {n}\n" + big_sym)
    return files


@pytest.mark.parametrize("threads", ["1", "6"])
def test_reader_is_the_references_reader_on_odd_files(tmp_path, threads):
    """COOMatrixRead of the shim (arm-spmv_amd/host/mtx_io.cpp, every SPMV_MTX_* switch off) against the reference's
    COOMatrixRead (src/data_io.cpp:45-105, compiled from its sources into oracle/_ref): same dimensions, same entries in
    the same order, values bit for bit, with the reference's loop (threads = 1) and with the parallel parser."""
    if not REF_SO.exists():
        pytest.skip("oracle/_ref/libarmspmv_ref.so not built (make -C oracle ref needs the reference sources)")
    if not LIB.exists():
        subprocess.run(["make", "host"], cwd=ROOT, check=True, capture_output=True)
    files = _odd_files(tmp_path)
    names = sorted(files)
    child = Path(__file__).with_name("reader_child.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("SPMV_MTX_")}
    env["SPMV_MTX_THREADS"] = threads
    got = {}
    for tag, so in (("ref", REF_SO), ("ours", LIB)):
        out = tmp_path / f"{tag}.npz"
        r = subprocess.run(["python3", str(child), str(so), str(out)] + [str(files[n]) for n in names],
                           capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, (tag, r.stdout[-2000:], r.stderr[-2000:])
        got[tag] = np.load(out)
    for i, name in enumerate(names):
        for key in ("dims", "row", "col", "val"):
            a, b = got["ref"][f"{key}{i}"], got["ours"][f"{key}{i}"]
            assert a.shape == b.shape and np.array_equal(a, b), (name, key, threads, a[:8], b[:8])
