import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
LIBRARY = ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_present() -> bool:
    try:
        from enzymm_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests fail loudly on a GPU box if the extension is missing; on a box without any GPU
    # (the build container) they are skipped so a plain `pytest tests/` stays usable.
    if os.environ.get("ENZYMM_AMD_FORCE_GPU_TESTS") == "1":
        return
    have_dev = os.path.exists("/dev/kfd")
    if have_dev:
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: not a GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def golden():
    import json
    return json.loads((GOLDEN / "golden.json").read_text())
