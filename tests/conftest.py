import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionfinish(session, exitstatus):
    """The per-variable parity maxima observed in this session (tests/nets_cases.py) -> one JSON under
    gpurun_out/ (merged back from the GPU box by gpurun; the judged copy is committed under profiles/)."""
    import json
    try:
        from tests.nets_cases import MARGINS
    except Exception:  # noqa: BLE001
        return
    if not MARGINS:
        return
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_margins.json'), 'w') as f:
            json.dump({'norm': 'per variable: max|got - oracle| / max|oracle|', 'bar': 1e-5,
                       'observed_max': dict(sorted(MARGINS.items()))}, f, indent=1)
    except OSError:
        pass
