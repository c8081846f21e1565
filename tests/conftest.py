import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that hangs (a kernel that never ends, a rank waiting in a collective) must become a FAILURE within minutes,
    not eat the whole run: every -m gpu test gets a time limit when pytest-timeout is installed (it is in this image)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") is not None and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900 if "bench" in item.name else 400))


@pytest.fixture(scope="session")
def gpu_ctx():
    from longtr_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
