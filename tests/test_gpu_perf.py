"""Wall-clock expectations on a real MI355X: marker `gpu_perf`, NOT part of `-m gpu` (the driver's parity run must not stop on a
clock).  Run with `pytest -m "gpu or gpu_perf"` (tools/env_sweeps.sh); under that expression the perf_expect() calls inside the
parity tests are assertions too."""
import numpy as np
import pytest

from conftest import perf_asserts_enabled

pytestmark = pytest.mark.gpu_perf


def test_perf_asserts_are_on_in_this_run():
    assert perf_asserts_enabled()


def test_c1_products_take_microseconds_and_auto_is_not_behind_any_forced_kernel(pkg):
    """BASELINE configs[0]'s shape resident on the device: the kernel AUTO keeps is within 15 % of the best forced one"""
    capi, synth = pkg.capi, pkg.synth
    ctx = capi.Context(0)
    n, k = 10_000, 16
    rp, cc, cv = synth.csr_uniform(0, n, n, k, seed=2024)
    A = ctx.csr(n, n, rp, cc, cv)
    x, y = ctx.vector_from(synth.vec_uniform(n, seed=2024)), ctx.vector(n)
    y.fill(0.0)
    ctx.apply_timed(A, x, y, 50)
    t_auto = min(ctx.apply_timed(A, x, y, 200) for _ in range(3))
    best = 1e9
    for kernel in (capi.CSR_VECTOR, capi.CSR_SCALAR, capi.CSR_PANEL):
        A.set_kernel(kernel)
        ctx.apply_timed(A, x, y, 50)
        best = min(best, min(ctx.apply_timed(A, x, y, 200) for _ in range(3)))
    assert t_auto < 0.02 and t_auto <= 1.15 * best + 0.0005, (t_auto, best)
    ctx.close()


def test_c3_band_runs_above_five_terabytes_per_second_of_the_bytes_it_moves(pkg):
    """BASELINE configs[2] at full size: N = 4M, K = 64, circulant band - 2.1 GB moved (8 bytes per slot, x, y twice)"""
    capi = pkg.capi
    ctx = capi.Context(0)
    n, k = 4_000_000, 64
    E = ctx.gen_ell_banded(n, n, k, seed=1)
    x, y = ctx.gen_vector(n, seed=1), ctx.vector(n)
    y.fill(0.0)
    ctx.apply_timed(E, x, y, 5)
    ms = min(ctx.apply_timed(E, x, y, 20) for _ in range(3))
    moved = 8.0 * n * k + 8.0 * n + 16.0 * n
    assert moved / (ms * 1e-3) / 1e12 >= 5.0, ms
    ctx.close()
