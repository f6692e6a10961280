"""Keyword routing helpers (pygrank/core/utils/__init__.py:11-60): constructor kwargs of a filter are split
between the preprocessor and the ConvergenceManager by signature inspection."""
import inspect


def call(method, kwargs, args=None):                         # utils/__init__.py:11-32
    if args:
        kwargs = dict(kwargs)
        for arg, val in zip(list(inspect.signature(method).parameters)[:len(args)], args):
            if arg in kwargs:
                raise Exception("Repeated argument to method " + method.__name__ + ": " + arg)
            kwargs[arg] = val
    accepted = inspect.signature(method).parameters
    return method(**{k: kwargs[k] for k in accepted if k in kwargs})


def remove_used_args(method, kwargs, args=None):             # utils/__init__.py:35-41
    if args:
        kwargs = dict(kwargs)
        for arg, val in zip(list(inspect.signature(method).parameters)[:len(args)], args):
            kwargs[arg] = val
    params = set(inspect.signature(method).parameters)
    return {k: v for k, v in kwargs.items() if k not in params}


def ensure_used_args(kwargs, methods=None):                  # utils/__init__.py:44-60
    known = []
    for method in (methods or []):
        known.extend(inspect.signature(method).parameters.keys())
    missing = set(kwargs.keys()) - set(known)
    if missing:
        raise Exception("No usage of argument(s) " + str(missing) + " found")
