import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import unina_yolo_dla_amd as u
    return u


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sd7(pkg):
    """Seed-7 synthetic state_dict of graph (A) (the weights every golden fixture was made with)."""
    return pkg.synth.make_state_dict(7, pkg.graph.Graph())


@pytest.fixture(scope="session")
def oracle_sd7(oracle_mod, sd7):
    h = oracle_mod.StateDict(sd7)
    yield h
    h.close()


def load_golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)
