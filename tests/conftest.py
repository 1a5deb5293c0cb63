import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def curve_batches(g):
    """The batches of a `curve*` fixture (tests/golden/make_golden.py: curve_case): seeded inputs, the
    target a fixed function of the input."""
    import torch
    cin, cout = int(g["meta.cin"]), int(g["meta.cout"])
    b = int(g["meta.batch"]); h, w = (int(v) for v in g["meta.hw"])
    out = []
    for i in range(int(g["meta.n_batches"])):
        gen = torch.Generator().manual_seed(100 + i)
        X = torch.rand(b, cin, h, w, generator=gen)
        out.append((X, X[:, :cout] * 0.5 + 0.25 * X[:, -cout:].flip(-1)))
    return out
