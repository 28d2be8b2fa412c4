import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# The banded pipeline picks its fill kernel by sub-batch size (K3, one pair per wavefront, up to 2048 pairs; K3v2, eight
# pairs per wavefront, above).  The tests' batches are small: pin K3v2 so that they keep exercising the kernel large
# batches run; tests/test_gpu_banded.py covers K3 (option 1) and the selection by size (option 0) explicitly.
os.environ.setdefault("BG_BAND_FILL_V1", "-1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """Some GPU tests hand torch device tensors to the engine.  torch bundles its own HIP runtime; it must
    initialise before libbiogpu's (the system one) has claimed the device with gigabytes of scratch, or its
    lazy init reports "No HIP GPUs are available" — so initialise it up front whenever a GPU is visible."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001 - CPU-only sessions
        pass
    yield
