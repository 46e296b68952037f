import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deprecated-lame-mirror_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import helpers
    return helpers.Oracle()


@pytest.fixture(scope="session")
def reference():
    import helpers
    if not helpers.have_reference():
        pytest.skip("oracle/_ref not built in this tree (needs /root/reference at build time)")
    return helpers.Reference()
