"""Small deterministic test matrices (inputs only).  Shared by tests/golden/make_golden.py and the tests.

The cases follow SURVEY.md 8c: (1) 8x8 tridiagonal, (2) 64x48 rectangular, unsorted, with duplicate entries
and empty rows, (3) one 4096-entry row among short rows (long-row / carry-out path), (4) BASELINE config 1:
10k x 10k, 16 entries per row, drawn from the engine's own counter-based generator.
"""
from __future__ import annotations

import hashlib

import numpy as np

from __graft_entry__ import load_package

synth = load_package().synth


def tri8():
    n = 8
    row, col, val = [], [], []
    for i in range(n):
        for j in (i - 1, i, i + 1):
            if 0 <= j < n:
                row.append(i)
                col.append(j)
                val.append(2.0 + 0.125 * i if i == j else -1.0 + 0.0625 * (i + j))
    x = np.arange(1, n + 1, dtype=np.float64) / 8.0
    return dict(name="tri8", nrow=n, ncol=n, row=np.array(row, np.int32), col=np.array(col, np.int32),
                val=np.array(val, np.float64), x=x)


def rect64x48():
    """unsorted COO, duplicates (same (i,j) several times), rows 5, 17, 40..44 and 63 empty"""
    rng = np.random.RandomState(20240607)
    nrow, ncol = 64, 48
    empty = {5, 17, 40, 41, 42, 43, 44, 63}
    rows = [r for r in range(nrow) if r not in empty]
    row = rng.choice(rows, size=400).astype(np.int32)
    col = rng.randint(0, ncol, size=400).astype(np.int32)
    val = rng.uniform(-1.0, 1.0, size=400)
    # force duplicates: repeat the first 40 coordinates with new values
    row = np.concatenate([row, row[:40]])
    col = np.concatenate([col, col[:40]])
    val = np.concatenate([val, rng.uniform(-1.0, 1.0, size=40)])
    perm = rng.permutation(row.size)
    x = rng.uniform(0.0, 1.0, size=ncol)
    return dict(name="rect64x48", nrow=nrow, ncol=ncol, row=row[perm], col=col[perm], val=val[perm], x=x)


def longrow():
    """300 x 5000, row 137 holds 4096 entries, the others 0..9; row-sorted"""
    rng = np.random.RandomState(7)
    nrow, ncol = 300, 5000
    lens = rng.randint(0, 10, size=nrow)
    lens[137] = 4096
    row = np.repeat(np.arange(nrow, dtype=np.int32), lens)
    col = rng.randint(0, ncol, size=row.size).astype(np.int32)
    val = rng.uniform(-1.0, 1.0, size=row.size)
    x = rng.uniform(0.0, 1.0, size=ncol)
    return dict(name="longrow", nrow=nrow, ncol=ncol, row=row, col=col, val=val, x=x)


C1_SEED = 2024


def c1():
    """BASELINE.json config 1: 10k x 10k, 16 entries per row, uniform columns, values U(-1,1), x U(0,1)"""
    n, k = 10_000, 16
    row_ptr, col, val = synth.csr_uniform(0, n, n, k, band=0, seed=C1_SEED)
    row = np.repeat(np.arange(n, dtype=np.int32), k)
    x = synth.vec_uniform(n, seed=C1_SEED)
    return dict(name="c1", nrow=n, ncol=n, row=row, col=col, val=val, x=x)


SMALL_CASES = (tri8, rect64x48, longrow)
ALL_CASES = SMALL_CASES + (c1,)


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
