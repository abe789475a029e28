"""numpy twin of the device generators in csrc/generate.hip.

draw(seed, stream, index) = splitmix64(key(seed, stream) + index): every element depends only on its
global index, so any row range can be rebuilt on the host bit for bit — used by the parity tests
(device generator vs this file) and by bench.py to build the CPU-baseline sample of the benchmark
matrix without copying it back from the GPU.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
STREAM_COL, STREAM_VAL, STREAM_VEC, STREAM_LEN = 1, 2, 3, 4


def splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stream_key(seed: int, stream: int) -> np.uint64:
    s = (int(seed) ^ ((0x9E3779B97F4A7C15 * (stream + 1)) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    return splitmix64(np.array([s], dtype=np.uint64))[0]


def _draw(key: np.uint64, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return splitmix64(idx.astype(np.uint64) + key)


def to_unit(r: np.ndarray) -> np.ndarray:  # [0,1)
    return (r >> np.uint64(11)).astype(np.float64) * 2.0**-53


def to_sym(r: np.ndarray) -> np.ndarray:  # [-1,1)
    return (r >> np.uint64(11)).astype(np.float64) * 2.0**-52 - 1.0


def to_range(r: np.ndarray, n: int) -> np.ndarray:  # [0,n)
    with np.errstate(over="ignore"):
        return (((r >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def vec_uniform(n: int, index_offset: int = 0, seed: int = 1) -> np.ndarray:
    idx = np.arange(index_offset, index_offset + n, dtype=np.uint64)
    return to_unit(_draw(stream_key(seed, STREAM_VEC), idx))


def csr_uniform(row_begin: int, row_end: int, ncol: int, k: int, band: int = 0, seed: int = 1):
    """rows [row_begin,row_end) with exactly k entries each -> (row_ptr int32, col int32, val fp64)"""
    nrow = row_end - row_begin
    grow = np.repeat(np.arange(row_begin, row_end, dtype=np.uint64), k)
    slot = np.tile(np.arange(k, dtype=np.uint64), nrow)
    gidx = grow * np.uint64(k) + slot
    rc = _draw(stream_key(seed, STREAM_COL), gidx)
    if band <= 0:
        col = to_range(rc, ncol)
    else:
        col = (grow.astype(np.int64) % ncol + to_range(rc, band) - band // 2) % ncol
    val = to_sym(_draw(stream_key(seed, STREAM_VAL), gidx))
    row_ptr = (np.arange(nrow + 1, dtype=np.int64) * k).astype(np.int32)
    return row_ptr, col.astype(np.int32), val


def ell_banded(nrow: int, ncol: int, k: int, seed: int = 1):
    """column-major ELL: (row i, slot d) at i + d*nrow, col = (i + d - k//2) mod ncol"""
    e = np.arange(nrow * k, dtype=np.int64)
    i = e % nrow
    d = e // nrow
    col = (i + d - k // 2) % ncol
    val = to_sym(_draw(stream_key(seed, STREAM_VAL), (i * k + d).astype(np.uint64)))
    return col.astype(np.int32), val


def powerlaw_lengths(nrow: int, max_len: int = 4096, seed: int = 1, sorted_by_length: bool = False) -> np.ndarray:
    """min(max_len, floor(8 / u)); u drawn from U(0,1], or (sorted_by_length) the quantiles u_i = (i + 1) / nrow: the same
    distribution with the rows sorted by length, the longest first"""
    if sorted_by_length:
        u = np.arange(1, nrow + 1, dtype=np.float64) / np.float64(nrow)
    else:
        u = 1.0 - to_unit(_draw(stream_key(seed, STREAM_LEN), np.arange(nrow, dtype=np.uint64)))
    q = np.floor(8.0 / u)
    return np.where(q >= max_len, max_len, q).astype(np.int32)


def coo_powerlaw(nrow: int, ncol: int, max_len: int = 4096, seed: int = 1, sorted_by_length: bool = False):
    """row-sorted COO with power-law row lengths -> (row, col, val)"""
    ln = powerlaw_lengths(nrow, max_len, seed, sorted_by_length).astype(np.int64)
    row = np.repeat(np.arange(nrow, dtype=np.int64), ln)
    start = np.concatenate(([0], np.cumsum(ln)))[:-1]
    s = np.arange(row.size, dtype=np.int64) - np.repeat(start, ln)
    gidx = (row * max_len + s).astype(np.uint64)
    col = to_range(_draw(stream_key(seed, STREAM_COL), gidx), ncol)
    val = to_sym(_draw(stream_key(seed, STREAM_VAL), gidx))
    return row.astype(np.int32), col.astype(np.int32), val
