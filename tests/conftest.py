import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libsbq.so normally arrives prebuilt (__graft_entry__.build()); if this checkout has none,
    build it once -- hipcc is on the CPU container and on the GPU box alike.  No fallback: if the
    build fails, every test that touches the library fails."""
    from sparsebit_amd import build as sbq_build

    if not os.path.exists(sbq_build.LIB):
        sbq_build.build(verbose=False)


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the real reference (tests/golden/gen_golden.py)."""
    path = os.path.join(ROOT, "tests", "golden", "ref_golden.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O


def golden_cases(prefix):
    path = os.path.join(ROOT, "tests", "golden", "ref_golden.npz")
    z = np.load(path, allow_pickle=False)
    return [c for c in z["cases"].tolist() if c.startswith(prefix)]


def rel_err(a, b):
    """max |a-b| / |b| with exact zeros required to match exactly (signed zero ignored)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    nz = b != 0
    err = 0.0
    if nz.any():
        err = float(np.max(np.abs(a[nz] - b[nz]) / np.abs(b[nz])))
    if (~nz).any():
        err = max(err, float(np.max(np.abs(a[~nz]))) * 1e30)  # any nonzero where 0 expected fails
    return err
