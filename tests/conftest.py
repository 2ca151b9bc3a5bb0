"""pytest configuration: markers and the path setup shared by every test.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol/load checks (no GPU needed).
`-m gpu`       : parity of the HIP path (through the C-ABI) against the oracle and the goldens.
"""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on the CPU")


@pytest.fixture(scope="session")
def golden_hashes():
    import json
    return json.loads((ROOT / "tests" / "golden" / "golden_hashes.json").read_text())


@pytest.fixture(autouse=True)
def _reset_library_options():
    """Run-time options set through xee.set_default_option are process-wide: put them back after every test."""
    yield
    try:
        import xee_pkg
        xee = xee_pkg.load()
        if xee.lib_path().exists():
            xee.reset_default_options()
    except Exception:
        pass
