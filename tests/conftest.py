import os
import sys

import numpy
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope='session')
def golden():
    path = os.path.join(ROOT, 'tests', 'golden', 'reference_outputs.npz')
    return numpy.load(path)


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope='session')
def oracle_keys(orc):
    """Full-size key set from the oracle's RNG-order-faithful key generation, seed 123
    (SURVEY §8d).  Returns (lwe_key, tlwe_key, CloudKeyArrays)."""
    rng = orc.DeterministicRNG(123)
    return orc.make_key_pair(rng)
