import os
import sys

import pytest

# PyTorch's wheel bundles its own HIP runtime under the system runtime's soname.  Loaded FIRST, it
# is the one libpk_mi355.so binds to as well -- one runtime in the process, the arrangement
# bench.py runs in; loaded after libpk_mi355.so, the process would hold two runtimes (the tests
# that use torch.distributed would still pass, RCCL initialisation in such a process does not).
try:
    import torch  # noqa: F401
except ImportError:
    pass

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
