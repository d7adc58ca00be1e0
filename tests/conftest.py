import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import isehr_amd  # noqa: E402,F401  registers the alias for the hyphenated package dir

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _keep_buffers_mode():
    """ISEHR_KEEP_BUFFERS=0 python -m pytest -m gpu ...: the whole suite with the spare-buffer slots switched off (every handle
    allocates and frees its own memory, as before round 5).  Default: the library's default (slots on)."""
    mode = os.environ.get("ISEHR_KEEP_BUFFERS")
    if mode is not None:
        from isehr_amd import _lib
        _lib.set_global_option("keep_buffers", int(mode))
    yield
