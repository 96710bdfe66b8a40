import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-minute CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a ROCm device: where none is visible they are skipped (instead of erroring), so that a plain `pytest tests` on
    a CPU host is green.  With a device present nothing is skipped -- a missing HIP library then fails loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="gpu test: no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
