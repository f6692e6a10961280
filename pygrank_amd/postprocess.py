"""Postprocessors that act on rank vectors already resident in HBM.

``Tautology`` is the default ``personalization_transform`` of every filter (abstract_filters.py:37) and ``Normalize`` is
the callable form of ``use_quotient`` (abstract_filters.py:131-132; the reference's filter tests compare outcomes after
``Normalize``, tests/test_filters.py:41-82).  ``Ordinals`` / ``Top`` / ``Threshold`` / ``Transformer`` / ``Sweep`` /
``LinearSweep`` are SURVEY.md 8f-3: elementwise kernels, device reductions and one device radix sort instead of the
reference's per-node Python dictionaries.  Behaviour follows pygrank/algorithms/postprocess/postprocess.py:7-80,106-290,
353-450.  The remaining postprocessors (oversampling, fairness, subgraph extraction) re-invoke the hot path and stay out of
scope."""
from pygrank_amd import backend
from pygrank_amd.device import DeviceVector
from pygrank_amd.signals import NodeRanking, to_signal
from pygrank_amd.utils import call, ensure_used_args, remove_used_args


class Postprocessor(NodeRanking):
    """A ranker wrapped around another ranker: ``rank`` runs the inner one and passes its outcome through ``_transform``."""

    def __init__(self, ranker=None):
        self.ranker = ranker

    def _transform(self, ranks, **kwargs):
        raise Exception("_transform method not implemented for the class " + type(self).__name__)

    def _apply(self, ranks, kwargs):
        return to_signal(ranks, call(self._transform, kwargs, [ranks]))

    def transform(self, ranks, *args, **kwargs):
        return self._apply(ranks, kwargs)

    def rank(self, *args, **kwargs):
        inner = self.ranker.rank(*args, **kwargs)
        return self._apply(inner, remove_used_args(self.ranker.rank, kwargs))      # keywords the inner ranker did not take

    def __lshift__(self, ranker):
        if isinstance(ranker, NodeRanking):
            self.ranker = ranker
            return ranker
        raise Exception("only a ranker can be shifted into a postprocessor, got " + type(ranker).__name__)

    # self-description: the wrapped algorithm's parts, then this step (postprocess.py: references() of every postprocessor
    # extends the inner ranker's list)
    def _reference(self):
        return type(self).__name__.lower() + " postprocessing"

    def references(self):
        inner = list(self.ranker.references()) if self.ranker is not None else []
        return inner + [self._reference()]

    # a wrapped filter's collaborators show through, so that ``postprocessor.convergence`` works like the filter's
    preprocessor = property(lambda self: self.ranker.preprocessor)
    convergence = property(lambda self: self.ranker.convergence,
                           lambda self, value: setattr(self.ranker, "convergence", value))


class Tautology(Postprocessor):
    """Changes nothing; without an inner ranker ``rank`` turns its input into a signal."""

    def references(self):
        return list(self.ranker.references()) if self.ranker is not None else ["tautology"]

    def transform(self, ranks, *args, **kwargs):
        return ranks

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        inner = self.ranker
        return to_signal(graph, personalization) if inner is None else inner.rank(graph, personalization, *args, **kwargs)


def _is_ranker(obj):
    return callable(getattr(obj, "rank", None))


class _NoOptions(Postprocessor):
    """Postprocessors whose transformation takes no keyword of its own: whatever is still there when the call arrives was
    meant for nobody (the reference's ``ensure_used_args`` at the top of every ``_transform``); the work is in ``_values``."""

    def _transform(self, ranks, **leftover):
        ensure_used_args(leftover)
        return self._values(ranks)


class Normalize(_NoOptions):
    """Rescales ranks by their maximum ("max", default), sum ("sum"), Euclidean norm ("L2"), or onto [0, 1] ("range").
    ``Normalize("sum", ranker)`` and ``Normalize(ranker, "sum")`` are the same thing (postprocess.py:124-131)."""

    _METHODS = ("max", "sum", "L2", "range")

    def __init__(self, ranker=None, method="max"):
        if ranker is not None and not _is_ranker(ranker):          # the method came first
            ranker, method = (method if _is_ranker(method) else None), ranker
        super().__init__(ranker if ranker is not None else Tautology())
        self.method = method

    def _values(self, ranks):
        how = self.method
        if how not in self._METHODS:
            raise Exception("Normalize: method must be one of " + ", ".join(self._METHODS) + ", not " + repr(how))
        x = ranks.np
        low = float(backend.min(x)) if how == "range" else 0.0
        high = float({"sum": backend.sum, "L2": lambda v: float(backend.dot(v, v)) ** 0.5}.get(how, backend.max)(x))
        if high == low:
            return ranks                                           # constant (or zero) signals stay as they are
        return (x - low) / (high - low)


def _device(ranks):
    x = ranks.np
    return x if isinstance(x, DeviceVector) else backend.to_array(x)


def _swap(first, second):
    """The reference's constructors accept (ranker, value) in either order (postprocess.py:124-131,246-252)."""
    if first is not None and not _is_ranker(first):
        first, second = (second if _is_ranker(second) else None), first
    return (first if first is not None else Tautology()), second


class Ordinals(_NoOptions):
    """1 for the highest rank, 2 for the second highest, ... (postprocess.py:163-195); ties keep node order."""

    def __init__(self, ranker=None):
        super().__init__(_swap(ranker, None)[0])

    def _values(self, ranks):
        return _device(ranks).ordinals()


class Transformer(_NoOptions):
    """Element-by-element expression, backend.exp by default (postprocess.py:198-243)."""

    def __init__(self, ranker=None, expr=None):
        ranker, expr = _swap(ranker, expr)
        super().__init__(ranker)
        self.expr = backend.exp if expr is None else expr

    def _values(self, ranks):
        return self.expr(ranks.np)


class Top(_NoOptions):
    """1 for the top-scored nodes, 0 for the rest; ``fraction_of_training`` >= 1 counts nodes, < 1 is a share of the graph
    (postprocess.py:246-290: every node that ties with the last kept one is kept too)."""

    def __init__(self, ranker=None, fraction_of_training=1):
        ranker, fraction = _swap(ranker, fraction_of_training)
        super().__init__(ranker)
        self.fraction_of_training = 1 if fraction is None else fraction

    def _values(self, ranks):
        x = _device(ranks)
        keep = self.fraction_of_training
        keep = int(keep * len(x)) if keep < 1 else int(keep)
        threshold = x.kth_largest(keep) if 1 <= keep <= len(x) else 0     # the reference's loop leaves 0 otherwise
        return (x >= threshold) * 1.0


class Threshold(_NoOptions):
    """1 above a threshold, 0 elsewhere (postprocess.py:293-350); "gap" = the score after the largest relative drop."""

    def __init__(self, ranker=None, threshold=0, inclusive=False):
        ranker, threshold = _swap(ranker, threshold)
        super().__init__(ranker)
        self.threshold = 0 if threshold is None else threshold
        self.inclusive = inclusive

    def _values(self, ranks):
        x = _device(ranks)
        # "gap": one device sort + two reductions (pgh_vec_gap_threshold)
        cut = x.gap_threshold() if self.threshold == "gap" else self.threshold
        return ((x >= cut) if self.inclusive else (x > cut)) * 1.0


class _UniformBaseline(_NoOptions):
    """Sweep family: personalized ranks set against the same ranker's non-personalized outcome, computed once per graph."""

    def __init__(self, ranker=None, uniform_ranker=None):
        super().__init__(ranker)
        self.uniform_ranker = uniform_ranker if uniform_ranker is not None else ranker
        self._baseline = {}

    def _uniforms(self, ranks):
        key = id(ranks.graph)
        if key not in self._baseline:
            self._baseline[key] = (ranks.graph, _device(self.uniform_ranker.rank(ranks.graph)))   # keeps the graph alive: ids are not reused
        return self._baseline[key][1]

    def __lshift__(self, ranker):
        self.uniform_ranker = super().__lshift__(ranker)
        return self.uniform_ranker


class Sweep(_UniformBaseline):
    """ranks / (1e-12 + uniform ranks) (postprocess.py:353-404)."""

    def _values(self, ranks):
        return _device(ranks) / (self._uniforms(ranks) + 1.E-12)


class LinearSweep(_UniformBaseline):
    """ranks - uniform ranks (postprocess.py:407-450)."""

    def _values(self, ranks):
        return _device(ranks) - self._uniforms(ranks)
