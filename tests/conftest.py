"""Shared pytest configuration: the `gpu` marker and repo-root imports."""
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu`)")


@pytest.fixture(scope="session")
def golden_dir() -> pathlib.Path:
    return GOLDEN
