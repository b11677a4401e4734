import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
SPEC_FIELDS = ("Q", "q", "Qf", "qf", "P", "R", "r", "A", "B", "V", "F", "W")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    actor = {k: g["actor_" + k] for k in SPEC_FIELDS}
    dyn = {k: g["dyn_" + k] for k in SPEC_FIELDS}
    return g, actor, dyn


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return float(np.abs(a - b).max() / scale)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle as OC
    OC.build()
    return OC
