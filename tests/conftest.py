import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "bayes-bridge_amd"),
             os.path.join(ROOT, "tests", "golden")):
    if path not in sys.path:
        sys.path.insert(0, path)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line(
        "markers",
        "needs_reference: imports /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import ref_import
    if ref_import.reference_available():
        return
    skip = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "needs_reference" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
