import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the reference's `omp atomic` loops (COO, CSC) are order-dependent: pin the oracle/_ref runs to one thread
os.environ.setdefault("OMP_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib

    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ctx(pkg):
    """one engine context on GPU 0; fails loudly (no skip, no fallback) if the HIP library or GPU is missing"""
    c = pkg.capi.Context(0)
    yield c
    c.close()


def golden(name):
    import numpy as np

    return np.load(ROOT / "tests" / "golden" / f"{name}.npz", allow_pickle=False)
