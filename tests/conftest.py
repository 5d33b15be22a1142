"""pytest config: `gpu` marker = needs a real MI355X (run by the driver with -m gpu on the GPU box)."""

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU (HIP kernels through the C ABI)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_ref

    cpu_ref.build()
    return cpu_ref


@pytest.fixture(scope="session")
def native_built():
    import __graft_entry__ as g

    if not g.LIB.exists():
        g.build()
    return g.LIB
