import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def amber():
    import amber_amd
    amber_amd.build_library()
    amber_amd.load_library()
    import oracle_binding
    oracle_binding.MATH_DEVICE = {amber_amd.MATH_GLIBC: oracle_binding.MATH_GLIBC,
                                  amber_amd.MATH_PORTABLE: oracle_binding.MATH_PORTABLE}[amber_amd.math_mode()]
    return amber_amd
