import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    """Keeps what the parity tests MEASURED (not only that they passed): gpurun_out/parity_record.json."""
    try:
        from util import PARITY
        if not PARITY:
            return
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_record.json"), "w") as f:
            json.dump({"exitstatus": int(exitstatus), "measured": PARITY}, f, indent=1, sort_keys=True, default=float)
    except Exception as e:      # a read-only tree must not fail the suite
        print("parity record not written: %s" % e)
