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
    config.addinivalue_line("markers", "slow: long-running test (GPU ones run only when the -m expression names `slow`)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` (the round-end run) leaves out the GPU tests that are also marked slow; `-m "gpu and slow"` runs them."""
    if 'slow' in (config.option.markexpr or ''):
        return
    skip = pytest.mark.skip(reason='gpu + slow: run with -m "gpu and slow"')
    for item in items:
        if 'gpu' in item.keywords and 'slow' in item.keywords:
            item.add_marker(skip)


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
