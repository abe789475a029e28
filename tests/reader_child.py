"""Child process of tests/test_host_io.py::test_reader_is_the_references_reader_on_odd_files: loads ONE library (the
engine's compat shim or the compiled reference, oracle/_ref), reads every file through the C++ symbol
COOMatrixRead(const char*, COOMatrix&) and stores what came back.  One library per process: both export the same
C++ names.

    python reader_child.py <library.so> <out.npz> <file> [<file> ...]
"""
import ctypes as C
import os
import sys

import numpy as np


class COO(C.Structure):  # reference include/matrix.h:9-16
    _fields_ = [("nrow", C.c_int), ("ncol", C.c_int), ("nnz", C.c_int), ("row_ind", C.POINTER(C.c_int)),
                ("col_ind", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_double))]


def main():
    lib = C.CDLL(sys.argv[1])
    read = getattr(lib, "_Z13COOMatrixReadPKcR9COOMatrix")
    read.argtypes = [C.c_char_p, C.POINTER(COO)]
    out = {}
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)  # both readers print their progress lines
    try:
        for i, path in enumerate(sys.argv[3:]):
            A = COO()
            read(path.encode(), C.byref(A))
            n = A.nnz
            out[f"dims{i}"] = np.array([A.nrow, A.ncol, A.nnz], dtype=np.int64)
            out[f"row{i}"] = np.ctypeslib.as_array(A.row_ind, (n,)).copy() if n else np.zeros(0, np.int32)
            out[f"col{i}"] = np.ctypeslib.as_array(A.col_ind, (n,)).copy() if n else np.zeros(0, np.int32)
            out[f"val{i}"] = np.ctypeslib.as_array(A.values, (n,)).copy().view(np.uint64) if n else np.zeros(0, np.uint64)
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved, 1)
    np.savez(sys.argv[2], **out)


if __name__ == "__main__":
    main()
