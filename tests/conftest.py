import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # HRF_TEST_LIB=tools/_build/libhrf_hip_<tag>.so: run the suite over a tuning variant of the library (make variant) before
    # its settings become the default build
    import os
    if os.environ.get("HRF_TEST_LIB"):
        import humanrf_amd._lib as hl
        hl.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["HRF_TEST_LIB"])


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        for item in items:   # a hung kernel / thread must fail one test, not eat the GPU budget of the whole run
            if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(300))
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
