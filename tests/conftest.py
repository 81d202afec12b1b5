import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not config.pluginmanager.hasplugin("timeout"):
        config.addinivalue_line("markers", "timeout(seconds): per-test limit (pytest-timeout)")


def pytest_collection_modifyitems(config, items):
    """A hung GPU test must fail, not stall the suite: every gpu test gets a 10-minute limit when pytest-timeout is
    present (it is in the image; without the plugin the marker is registered below and ignored)."""
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
                item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
