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
    # The boxes show 256 cores and grant 16: BLAS / OpenMP pools sized by the visible count (the oracle's float64 products, torch's
    # CPU ops) exhaust the cgroup's CPU quota within milliseconds and the whole process tree is frozen until the next period.
    # Cap them for this process and - through the environment - for every server / worker / bench process the tests start.
    from vod_amd.hostcpu import limit_cpu_threads

    n = limit_cpu_threads()
    try:
        import threadpoolctl

        config._vod_threadpool_limit = threadpoolctl.threadpool_limits(limits=n)  # BLAS libraries already loaded (NumPy)
    except ImportError:  # pragma: no cover
        pass


@pytest.fixture(scope="session")
def golden_dir() -> pathlib.Path:
    return GOLDEN
