"""
conftest.py for running the REFERENCE's own pytest files, unmodified, against the drop-in.

The reference's test/conftest.py (lines 80-127) builds its fixtures on Reikna threads of PyCUDA / PyOpenCL devices; this
one provides the same fixture names -- thread, key_pair, context, context_and_key_pair, transform_type,
heavy_performance_load -- and the same command-line options over a nufhe_amd DeviceThread.  tools/run_reference_tests.sh
copies test_api_high_level.py, test_api_low_level.py, test_gates.py and utils.py from /root/reference/test next to this
file (into the git-ignored tools/scratch/, which travels to the GPU box; the copies are never committed) and runs them with
the repo root on PYTHONPATH, so that `import nufhe` is the alias package.  `reikna/` beside this file holds the one name
test_gates.py imports from Reikna (cuda_id).
"""
import pytest

from nufhe import make_key_pair, DeterministicRNG, Context


def pytest_addoption(parser):
    parser.addoption("--heavy-performance-load", action="store_true", default=False)
    parser.addoption("--transform", action="store", default="all", choices=["NTT", "FFT", "all"])


def pytest_configure(config):
    config.addinivalue_line("markers", "perf: the reference's performance tests")


def pytest_generate_tests(metafunc):
    if 'transform_type' in metafunc.fixturenames:
        opt = metafunc.config.option.transform
        metafunc.parametrize("transform_type", ['NTT', 'FFT'] if opt == 'all' else [opt])


@pytest.fixture(scope='session')
def thread():
    from nufhe_amd.device import DeviceThread
    return DeviceThread(0)


@pytest.fixture(scope='session')
def heavy_performance_load(request):
    return request.config.option.heavy_performance_load


@pytest.fixture(scope='session')
def key_pair(thread):
    rng = DeterministicRNG()
    secret_key, cloud_key = make_key_pair(thread, rng)
    return secret_key, cloud_key


@pytest.fixture(scope='session')
def context(thread):
    return Context(thread=thread)


@pytest.fixture(scope='session')
def context_and_key_pair(context):
    secret_key, cloud_key = context.make_key_pair()
    return context, secret_key, cloud_key
