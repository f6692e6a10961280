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
_engines = {}            # name -> imported engine module
_loaded = None


def safe_div(nom, denom, default=0):
    """nom / denom, or `default` for a zero denominator (backend/__init__.py:14-17)."""
    return nom / denom if denom != 0 else default


def safe_inv(x):
    """Elementwise 1 / x with 1 / 0 = 0 (backend/__init__.py:20-23); one kernel on a device vector."""
    if hasattr(x, "_unary"):
        from pygrank_amd import _lib
        return x._unary(_lib.SAFE_INV)
    x = np.asarray(x, dtype=np.float64)
    return np.divide(1.0, x, out=np.zeros_like(x), where=x != 0)


def backend_name():
    return _loaded.backend_name() if _loaded is not None else "no backend loaded"


class Backend:
    """``with Backend(name):`` -- the named engine inside the block, the previous one (if any was loaded) afterwards
    (backend/__init__.py:26-37)."""

    def __init__(self, mod_name):
        self.mod_name, self._before = mod_name, None

    def __enter__(self):
        self._before = backend_name()
        return load_backend(self.mod_name)

    def __exit__(self, *exc):
        if self._before in SUPPORTED:
            load_backend(self._before)
        return False


def _primitive_of(arg):
    """What an engine function receives in place of a wrapper object: the dense vector of a signal, the matrix handle of an
    Adjacency (class names, not imports: signals / preprocessing import this package)."""
    kind = type(arg).__name__
    if kind == "GraphSignal":
        return arg.np
    if kind == "Adjacency" and hasattr(arg, "array"):
        return arg.array
    return arg


def _conv_over(engine_conv):
    """conv keeps the signal-ness of its first operand: signal in, signal (on the same graph) out."""
    def conv(x, M):
        out = engine_conv(_primitive_of(x), _primitive_of(M))
        if type(x).__name__ != "GraphSignal":
            return out
        from pygrank_amd.signals import to_signal
        return to_signal(x, out)
    return conv


def _exposed(function):
    if function.__name__ == "conv":
        return _conv_over(function)

    def exposed(*args, **kwargs):
        if kwargs:
            return function(*map(_primitive_of, args), **{name: _primitive_of(value) for name, value in kwargs.items()})
        return function(*map(_primitive_of, args))
    exposed.__name__, exposed.__doc__ = function.__name__, function.__doc__
    return exposed


def load_backend(mod_name):
    """Imports the engine (once), checks it against the 29-function contract (specification.API), initialises it and
    publishes the unwrapping versions of its functions on this package and on ``pygrank_amd`` (backend/__init__.py:40-84)."""
    global _loaded
    if mod_name not in SUPPORTED:
        raise Exception("Unsupported backend " + str(mod_name))
    engine = _engines.get(mod_name)
    if engine is None:
        engine = _engines[mod_name] = importlib.import_module("." + mod_name, __name__)
    missing = [api for api in specification.API if not hasattr(engine, api)]
    if missing:
        raise Exception("Missing implementation for " + missing[0])
    engine.backend_init()                 # raises without an MI355X / without the HIP library
    homes = [sys.modules[name] for name in (__name__, "pygrank_amd") if name in sys.modules]
    for api in specification.API:
        published = _exposed(getattr(engine, api))
        for home in homes:
            setattr(home, api, published)
    _loaded = engine
    return engine


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
