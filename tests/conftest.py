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


def model_digest(model):
    import hashlib

    h = hashlib.sha256()
    for k in sorted(model):
        h.update(np.ascontiguousarray(model[k]).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="session")
def synth_model():
    from smplpp_amd import model_io

    return model_io.synthetic_model()


@pytest.fixture(scope="session")
def oracle_synth(synth_model):
    """C oracle on the synthetic model, with the reference's unordered_map adjacency order (from the fixture)."""
    from oracle import cpu

    o = cpu.OracleModel(synth_model)
    g = np.load(os.path.join(GOLDEN, "ik_synth.npz"))
    adj = g["adjacency"]
    for v in range(o.V):
        row = adj[v]
        o.set_adjacency(v, row[row >= 0])
    return o


@pytest.fixture(scope="session")
def golden_fk_synth():
    return np.load(os.path.join(GOLDEN, "fk_synth.npz"))


@pytest.fixture(scope="session")
def golden_ik_synth():
    return np.load(os.path.join(GOLDEN, "ik_synth.npz"))
