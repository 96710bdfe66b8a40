"""CPU: the C-ABI shared library loads and exports every symbol declared in include/clipself_hip.h
(no compute calls -- there is no GPU in the build container)."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "clipself_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from clipself_amd import hip
    if not hip.library_path().exists():
        hip.build_library()
    lib = hip.load_library()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/clipself_hip.h but not exported"
    assert sorted(hip.SIGNATURES) == names, "hip.SIGNATURES and the header disagree"


def test_ops_refuse_to_run_without_gpu():
    import torch
    from clipself_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hip.HipOps()
