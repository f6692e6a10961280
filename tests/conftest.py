import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))


@pytest.fixture(scope="session")
def golden_norm():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_norm.npz"))


@pytest.fixture(scope="session")
def graphs():
    import cases
    cache = {}

    def get(key):
        if key not in cache:
            cache[key] = cases.GRAPHS[key]()
        return cache[key]
    return get


def _ensure_oracle_built():
    import subprocess
    build = os.environ.get("PGH_ORACLE_BUILD_DIR")          # `make -C oracle asan-test`: the sanitizer builds of the checkers
    if build:
        return build
    build = os.path.join(ROOT, "oracle", "_build")
    if not (os.path.exists(os.path.join(build, "libpgh_host_oracle.so"))
            and os.path.exists(os.path.join(build, "liboracle_spmv.so"))):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return build


@pytest.fixture(scope="session")
def oracle_build_dir():
    return _ensure_oracle_built()


@pytest.fixture()
def host_engine(oracle_build_dir):
    """Routes pygrank_amd's ctypes binding to the host test double (oracle/host_abi.cpp) for CPU tests of
    the host-side Python.  The product never does this."""
    import pygrank_amd as pg
    import host_double
    host_double.install(os.path.join(oracle_build_dir, "libpgh_host_oracle.so"))
    pg.load_backend("hip")
    yield pg
    host_double.remove()


@pytest.fixture(scope="session")
def gpu_engine():
    """The real engine: libpgh_hip.so on an MI355X (tests marked gpu)."""
    import pygrank_amd as pg
    from pygrank_amd import _lib
    pg.load_backend("hip")
    assert _lib.runtime_name().startswith("hip:")
    return pg
