import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def tables():
    from eigen_zeth_amd.poseidon_constants import default_round_constants, default_mds
    return (np.array(default_round_constants(), dtype=np.uint64), np.array(default_mds(), dtype=np.uint64))


@pytest.fixture(scope="session")
def prover():
    """one zp_ctx on GPU 0 for the whole session (fails loudly without the HIP library / a GPU)"""
    from eigen_zeth_amd.native import Prover
    p = Prover(0)
    yield p
    p.close()
