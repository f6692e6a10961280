"""Backend loader of the build (counterpart of pygrank/core/backend/__init__.py:26-133).

Same mechanics as the reference loader: an allow-list, lazy import of ``pygrank_amd.backend.<name>``, a
completeness check against the 29-function contract, wrappers that unwrap GraphSignal -> ``.np`` and
Adjacency-like -> ``.array`` (``conv`` re-wraps a GraphSignal result), and installation of the wrapped
functions on ``pygrank_amd`` and ``pygrank_amd.backend``.  The allow-list holds exactly one engine -- "hip":
this package accelerates the propagation path and has no CPU engine to fall back to.  Selection through the
``pygrankBackend`` environment variable is honoured (only "hip" is valid).
"""
import importlib
import os
import sys

import numpy as np

from pygrank_amd.backend import specification

SUPPORTED = ["hip"]
_imported_mods = dict()
_loaded = None


def safe_div(nom, denom, default=0):      # backend/__init__.py:14-17
    if denom == 0:
        return default
    return nom / denom


def safe_inv(x):                          # backend/__init__.py:20-23
    if hasattr(x, "_unary"):
        from pygrank_amd import _lib
        return x._unary(_lib.SAFE_INV)
    y = np.copy(x)
    y[x != 0] = 1. / x[x != 0]
    return y


class Backend:                            # backend/__init__.py:26-37
    def __init__(self, mod_name):
        self.mod_name = mod_name

    def __enter__(self):
        self._previous_backend = backend_name()
        load_backend(self.mod_name)
        return _imported_mods[self.mod_name]

    def __exit__(self, *args, **kwargs):
        if self._previous_backend in SUPPORTED:
            load_backend(self._previous_backend)
        return False


def backend_name():
    return _loaded.backend_name() if _loaded is not None else "no backend loaded"


def _unwrap(arg):
    if arg.__class__.__name__ == "GraphSignal":
        return arg.np
    if hasattr(arg, "array") and arg.__class__.__name__ == "Adjacency":
        return arg.array
    return arg


def _wrap(method):
    if method.__name__ == "conv":
        def conv(x, M):
            M = _unwrap(M)
            if x.__class__.__name__ == "GraphSignal":
                from pygrank_amd.signals import to_signal
                return to_signal(x, method(x.np, M))
            return method(x, M)
        return conv

    def converted(*args, **kwargs):
        return method(*[_unwrap(a) for a in args], **{k: _unwrap(v) for k, v in kwargs.items()})
    converted.__name__ = method.__name__
    converted.__doc__ = method.__doc__
    return converted


def load_backend(mod_name):               # backend/__init__.py:40-84
    global _loaded
    if mod_name not in SUPPORTED:
        raise Exception("Unsupported backend " + str(mod_name))
    if mod_name in _imported_mods:
        mod = _imported_mods[mod_name]
    else:
        mod = importlib.import_module("." + mod_name, __name__)
        _imported_mods[mod_name] = mod
    for api in specification.API:
        if api not in mod.__dict__:
            raise Exception("Missing implementation for " + str(api))
    mod.backend_init()                    # raises without an MI355X / without the HIP library
    targets = [sys.modules[__name__]]
    if "pygrank_amd" in sys.modules:
        targets.append(sys.modules["pygrank_amd"])
    for api in specification.API:
        wrapped = _wrap(mod.__dict__[api])
        for target in targets:
            setattr(target, api, wrapped)
    _loaded = mod
    return mod


def get_backend_preference():             # backend/__init__.py:87-109 (env only; no config file is written)
    name = os.environ.get("pygrankBackend", "hip")
    return name if name in SUPPORTED else "hip"


def _lazy(api):
    """Until a backend is loaded every contract function loads the preferred engine on first use."""
    def stub(*args, **kwargs):
        load_backend(get_backend_preference())
        return getattr(sys.modules[__name__], api)(*args, **kwargs)
    stub.__name__ = api
    return stub


for _api in specification.API:
    if _api != "backend_name":
        globals()[_api] = _lazy(_api)
