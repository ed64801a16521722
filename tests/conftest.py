import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def grut_lib():
    """The HIP library, built in-tree (hipcc cross-compiles without a GPU)."""
    import importlib
    build = importlib.import_module("3dgrut_amd.build")
    build.build()
    return importlib.import_module("3dgrut_amd._abi").load_library()
