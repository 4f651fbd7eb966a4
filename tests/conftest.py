import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def u16_to_f32(a: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32."""
    return (a.astype(np.uint32) << 16).view(np.float32)


@pytest.fixture(scope="session")
def ops_golden():
    return np.load(os.path.join(GOLDEN, "ops.npz"))


@pytest.fixture(scope="session")
def ss_golden():
    z = np.load(os.path.join(GOLDEN, "supersampler.npz"))
    sd = {str(k): u16_to_f32(z["sd." + str(k)]) for k in z["keys"]}
    return z, sd


@pytest.fixture(scope="session")
def reprfn_golden():
    z = np.load(os.path.join(GOLDEN, "reprfn.npz"))
    sd = {str(k): u16_to_f32(z["net.sd." + str(k)]) for k in z["net.keys"]}
    return z, sd
