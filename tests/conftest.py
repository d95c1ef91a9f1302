import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
# The suite runs on the LAB build: the product's sources compiled with -DAMBER_LAB, i.e. the same kernels plus the known-answer entry points,
# the signature instantiations and the measured-and-kept schedulers (include/amber_hip_lab.h).  tests/test_product_library.py checks that the
# product library exports only the documented ABI and renders the same bits.  (A measurement build may be selected from outside.)
os.environ.setdefault("AMBER_AMD_LIB", "libamber_hip_lab.so")
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
