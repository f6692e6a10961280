"""The C-ABI library builds, loads, and exports every symbol include/pgh.h declares (no compute calls: this
runs in the GPU-less container).  Also: the product path fails loudly without an MI355X."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "pygrank_amd", "csrc", "libpgh_hip.so")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pgh.h")).read()
    return sorted(set(re.findall(r"\b(pgh_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "pygrank_amd", "csrc")])
    return LIB


def test_header_and_binding_agree():
    from pygrank_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_hip_library_exports_every_declared_symbol(built_lib):
    from pygrank_amd import _lib
    cdll = _lib.load_library(built_lib)            # dlopen + bind; raises on a missing symbol
    for name in declared_symbols():
        assert hasattr(cdll, name)
    assert cdll.pgh_runtime_name().decode() == "hip:gfx950"


def test_library_contains_gfx950_code_object(built_lib):
    out = subprocess.run(["strings", "-n", "6", built_lib], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_test_double_exports_every_declared_symbol(oracle_build_dir):
    import ctypes
    cdll = ctypes.CDLL(os.path.join(oracle_build_dir, "libpgh_host_oracle.so"))
    for name in declared_symbols():
        assert hasattr(cdll, name)
    cdll.pgh_runtime_name.restype = ctypes.c_char_p
    assert cdll.pgh_runtime_name().decode() == "host-oracle"


def test_product_path_fails_loudly_without_gpu(built_lib):
    """No CPU fallback: without a visible MI355X the first backend call raises."""
    import ctypes
    from pygrank_amd import _lib
    count = ctypes.c_int(-1)
    cdll = _lib.load_library(built_lib)
    cdll.pgh_device_count(ctypes.byref(count))
    if count.value > 0:
        pytest.skip("a GPU is visible")
    code = ("import pygrank_amd as pg\n"
            "try:\n    pg.sum([1.0, 2.0])\n    print('NO-ERROR')\n"
            "except Exception as e:\n    print('RAISED', type(e).__name__)\n")
    out = subprocess.run(["python", "-c", code], capture_output=True, text=True, cwd=ROOT).stdout
    assert "RAISED EngineError" in out


def test_product_refuses_foreign_runtime(oracle_build_dir):
    """_lib.ensure_init only drives a library whose runtime name starts with "hip:" (tests widen ACCEPTED_RUNTIMES from the
    outside, tests/host_double.py; the product has no such switch)."""
    import ctypes
    from pygrank_amd import _lib
    saved = (_lib._lib, _lib._initialised, _lib.ACCEPTED_RUNTIMES)
    try:
        _lib._lib = _lib._bind(ctypes.CDLL(os.path.join(oracle_build_dir, "libpgh_host_oracle.so")))
        _lib._initialised = False
        _lib.ACCEPTED_RUNTIMES = ("hip:",)
        with pytest.raises(_lib.EngineError):
            _lib.ensure_init()
    finally:
        _lib._lib, _lib._initialised, _lib.ACCEPTED_RUNTIMES = saved
