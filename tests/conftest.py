import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """after every GPU test: drop the solver objects the test left behind and hand their blocks back to the device
    (a 512^3 case holds ~25 GB; dozens of cases in one process otherwise pile up until the box runs out).
    X3D_TEST_MEMLOG=<file>: append the free device memory after each test (diagnosis)."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            log = os.environ.get("X3D_TEST_MEMLOG")
            if log:
                free, total = torch.cuda.mem_get_info()
                with open(log, "a") as f:
                    f.write("%-110s free %7.1f GiB of %.1f\n" % (request.node.nodeid[-110:], free / 2 ** 30, total / 2 ** 30))
    except Exception:
        pass
