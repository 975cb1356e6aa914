import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def goldens():
    return np.load(os.path.join(GOLDEN_DIR, "reference_goldens.npz"))


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN_DIR, "reference_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def goldens_expand():
    return np.load(os.path.join(GOLDEN_DIR, "reference_goldens_expand.npz"))


@pytest.fixture(scope="session")
def golden_expand_meta():
    with open(os.path.join(GOLDEN_DIR, "reference_goldens_expand.json")) as f:
        return json.load(f)
