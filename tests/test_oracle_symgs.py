"""CPU test of the oracle's symmetric Gauss-Seidel sweep (oracle/spmv_oracle.c: orc_symgs) against a dense numpy
statement of the same definition.  The reference reserves `diagonal // for SymGS` fields (include/matrix.h:36,81) and
has no sweep: nothing of the reference's to pin this to ("parity unpinned"); the GPU tests compare the engine with
this function."""
import numpy as np

import oracle_lib as ol


def _dense_symgs(a, b, x, sweeps):
    n = len(b)
    for _ in range(sweeps):
        for order in (range(n), range(n - 1, -1, -1)):
            for i in order:
                x[i] = (b[i] - a[i] @ x + a[i, i] * x[i]) / a[i, i]
    return x


def test_orc_symgs_against_a_dense_sweep_and_its_error_return():
    orc = ol.load_oracle()
    rng = np.random.default_rng(3)
    n = 60
    a = np.where(rng.uniform(size=(n, n)) < 0.15, rng.uniform(-1, 1, (n, n)), 0.0)
    a[np.arange(n), np.arange(n)] = np.abs(a).sum(axis=1) + 1.0
    r, c = np.nonzero(a)
    rp = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int32)
    cc, cv = c.astype(np.int32), a[r, c]
    # the diagonal entry of row 5 split in two: duplicates are summed
    k = int(np.flatnonzero((r == 5) & (c == 5))[0])
    cc2, cv2 = np.insert(cc, k, 5).astype(np.int32), np.insert(cv, k, 0.25 * cv[k])
    cv2[k + 1] *= 0.75
    rp2 = rp.copy()
    rp2[6:] += 1
    b, x0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    for sweeps in (1, 3):
        want = _dense_symgs(a, b, x0.copy(), sweeps)
        for rp_, cc_, cv_ in ((rp, cc, cv), (rp2, cc2, cv2)):
            got = x0.copy()
            assert ol.symgs(orc, rp_, cc_, cv_, b, got, sweeps) == 0
            assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
    # an explicit order: the dense sweep over the permuted system
    perm = rng.permutation(n).astype(np.int32)
    ap = a[np.ix_(perm, perm)]
    want = np.empty(n)
    want[perm] = _dense_symgs(ap, b[perm], x0[perm].copy(), 2)
    got = x0.copy()
    assert ol.symgs(orc, rp, cc, cv, b, got, 2, order=perm) == 0
    assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))
    # the greedy colouring: proper on the symmetrised pattern, smallest-first, rows by (colour, row)
    sym = (a != 0) | (a.T != 0)
    rs, cs = np.nonzero(sym)
    rps = np.concatenate([[0], np.cumsum(np.bincount(rs, minlength=n))]).astype(np.int32)
    ncol, colour, order = ol.greedy_colour_order(orc, rps, cs.astype(np.int32))
    assert ncol == colour.max() + 1 and sorted(order) == list(range(n))
    assert all(colour[i] != colour[j] for i, j in zip(rs, cs) if i != j)
    assert all(c == 0 or any(sym[i, j] and j < i and colour[j] == c - 1 for j in range(n)) for i, c in enumerate(colour))
    assert np.array_equal(order, np.lexsort((np.arange(n), colour)))
    # many sweeps converge to the solution of a diagonally dominant system
    x = np.zeros(n)
    ol.symgs(orc, rp, cc, cv, b, x, 200)
    assert np.max(np.abs(a @ x - b)) <= 1e-12
    # a row without a diagonal entry is reported (1 + row) and the sweep stops there
    keep = ~((r == 7) & (c == 7))
    rp3 = np.concatenate([[0], np.cumsum(np.bincount(r[keep], minlength=n))]).astype(np.int32)
    x = x0.copy()
    assert ol.symgs(orc, rp3, cc[keep].astype(np.int32), cv[keep], b, x, 1) == 8
    assert np.array_equal(x[7:], x0[7:])
