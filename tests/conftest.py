import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TRUSTED_SETUP = os.path.join(GOLDEN, "trusted_setup_4096.json")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_setup():
    """Oracle `Setup` without the 4096 per-point subgroup checks (those are
    exercised on a sample in test_oracle_kat.py; the full pass costs ~15 s)."""
    from oracle.pyref.setup import Setup

    return Setup.load_json(TRUSTED_SETUP, subgroup_checks=False)
