import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.hookimpl(trylast=True)
def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # a stuck rendezvous of one of the world-2 CPU tests must fail that test, not hang the suite for torch's 30-minute default
    # (the gloo groups are created with a 3-minute timeout; with pytest-timeout present every test is also bounded: its own
    # configure has run by now and left None when no --timeout / ini value was given)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config, "_env_timeout", None):
        config._env_timeout = 420.0


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
