"""Keyword routing between a filter's constructor and the objects it builds.

A filter accepts one flat ``**kwargs`` and hands every keyword to whichever of its collaborators (the preprocessor,
the ConvergenceManager) declares a parameter of that name; a keyword nobody declares is an error.  Behaviour of
pygrank/core/utils/__init__.py:11-60 (``call``, ``remove_used_args``, ``ensure_used_args``), which the reference's filter
constructors rely on (abstract_filters.py:35-39)."""
import inspect


def _parameter_names(method):
    return list(inspect.signature(method).parameters)


def _with_positionals(method, kwargs, args):
    """kwargs extended by ``args`` bound to the leading parameters of ``method``; ``strict`` callers reject duplicates."""
    merged = dict(kwargs)
    bound = []
    if args:
        for name, value in zip(_parameter_names(method), args):
            bound.append(name)
            merged[name] = value
    return merged, bound


def call(method, kwargs, args=None):
    """``method(**subset)`` where subset = the entries of ``kwargs`` (plus positionals in ``args``) it has parameters for."""
    merged, bound = _with_positionals(method, kwargs, args)
    clash = [name for name in bound if name in kwargs]
    if clash:
        raise Exception("Repeated argument to method " + method.__name__ + ": " + clash[0])
    wanted = set(_parameter_names(method))
    return method(**{name: value for name, value in merged.items() if name in wanted})


def remove_used_args(method, kwargs, args=None):
    """The part of ``kwargs`` that ``method`` has no parameter for."""
    merged, _ = _with_positionals(method, kwargs, args)
    wanted = set(_parameter_names(method))
    return {name: value for name, value in merged.items() if name not in wanted}


def ensure_used_args(kwargs, methods=None):
    """Raises when some keyword is declared by none of ``methods``."""
    declared = set()
    for method in methods or ():
        declared.update(_parameter_names(method))
    unknown = set(kwargs) - declared
    if unknown:
        raise Exception("No usage of argument(s) " + str(unknown) + " found")
