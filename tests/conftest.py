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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def measured(what, got, ref, bound):
    """Max abs difference of two arrays, asserted against `bound` and appended to
    gpurun_out/parity_errors.log (test id, what, measured, bound) - the full-size parity tests assert
    a small multiple of what they measure, and this log is where those numbers come from."""
    d = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.log"), "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{what}\t{d:.4e}\t{bound:.4e}\n")
    except OSError:
        pass
    assert d <= bound, f"{what}: max abs diff {d:.3e} > {bound:.3e}"
    return d
