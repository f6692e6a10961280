"""TESTS ONLY: routes pygrank_amd's ctypes binding to the host restatement of the C-ABI (oracle/host_abi.cpp) so that the
host-side Python (signals, filters, convergence bookkeeping, the gloo row-partition path) can be exercised without a GPU.
The product knows nothing of this: the rebinding happens here, from the outside."""
import ctypes
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE = os.path.join(os.environ.get("PGH_ORACLE_BUILD_DIR") or os.path.join(ROOT, "oracle", "_build"), "libpgh_host_oracle.so")
_saved = None


def install(path=DOUBLE):
    global _saved
    from pygrank_amd import _lib
    if _saved is None:
        _saved = (_lib._lib, _lib._initialised, _lib.ACCEPTED_RUNTIMES)
    _lib._lib = _lib._bind(ctypes.CDLL(path))
    _lib._initialised = False
    _lib.ACCEPTED_RUNTIMES = _saved[2] + ("host-oracle",)


def remove():
    global _saved
    from pygrank_amd import _lib
    _lib._lib, _lib._initialised, _lib.ACCEPTED_RUNTIMES = None, False, (_saved[2] if _saved else ("hip:",))
    _saved = None


def active():
    from pygrank_amd import _lib
    return _lib._lib is not None and not _lib.runtime_name().startswith("hip:")
