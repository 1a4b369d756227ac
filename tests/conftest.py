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
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def golden_layers(g):
    layers = []
    i = 0
    while f"l{i}_W" in g:
        layers.append({k: g[f"l{i}_{k}"] for k in ("W", "gamma", "beta", "mean", "var")})
        i += 1
    return layers


READER_CASES = ["reader_nusc_b2", "reader_c2_b1", "reader_c1_b3_gap", "reader_waymo_b1", "reader_single_point"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O

    O.build()
    return O
