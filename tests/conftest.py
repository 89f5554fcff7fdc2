import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def zenv_mod():
    # torch (used by the interop / sharding tests) bundles its own libamdhip64: it must be the first
    # HIP runtime loaded so that libzenv_hip.so binds to the same copy (streams and pointers are shared)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import __graft_entry__ as g
    g.build()
    import combinatorial_rl_tasks_amd as Z
    return Z
