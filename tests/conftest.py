import os
import sys

import pytest

# PyTorch first: it brings its own copy of the HIP runtime, and a process must end up with one runtime.  Loaded
# before libptcore.so, torch's copy also serves the library (same soname); the other way round torch finds no GPU.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libptref.so (the compiled reference)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oraclebind
    if not os.path.exists(oraclebind.ORACLE_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libptoracle.so"])
    return oraclebind.load_oracle()


@pytest.fixture(scope="session")
def ref_lib():
    from oracle import oraclebind
    if not oraclebind.have_ref():
        pytest.skip("oracle/_ref/libptref.so not built (reference tree absent)")
    return oraclebind.load_ref()


@pytest.fixture(scope="session")
def core_lib():
    from pathtracer_amd import api
    return api.load_core()
