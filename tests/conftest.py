import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.path.exists("/dev/kfd"):
        # On a GPU box: tests that need helper processes fork them from a server that is started HERE, while this
        # process has not touched the GPU yet (a process that holds a GPU context must not exec another program).
        try:
            from multiprocessing import forkserver

            forkserver.ensure_running()
        except Exception:  # noqa: BLE001 -- the test that needs it will say so
            pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max-norm error relative to max|b| (the parity metric of SURVEY.md 8d)."""
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.max(np.abs(b))
    if scale == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / scale)


def l2_rel_err(a, b):
    """L2-relative error |a - b|_2 / |b|_2 -- the second parity gate of SURVEY.md 8d ("L2-relative < 1e-10 (fp64),
    field likewise").  float64 accumulation whatever the inputs' type."""
    a = np.asarray(a)
    b = np.asarray(b)
    d = (a - b).ravel()
    den = float(np.sqrt(np.vdot(b.ravel(), b.ravel()).real))
    num = float(np.sqrt(np.vdot(d, d).real))
    return num if den == 0.0 else num / den
