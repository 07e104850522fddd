import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One device context for the whole GPU session.  No skip when the GPU is
    missing: the product has no CPU fallback and -m gpu must fail loudly."""
    import torch  # noqa: F401  (loads the HIP runtime the library shares)
    from inplacemsdradixsort_amd import MsdContext
    c = MsdContext(0)
    yield c
    c.close()
