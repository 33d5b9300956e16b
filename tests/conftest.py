import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs through the HIP library)")
    # CPU-side libraries are cheap to (re)build; the HIP library is built by __graft_entry__.build()
    from bundleadjustment_amd import build
    build.build_host()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libba_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libba_oracle.so"], check=True)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "collinearity_golden.json")) as f:
        return json.load(f)


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def projection_golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "projection_golden.json")) as f:
        return json.load(f)
