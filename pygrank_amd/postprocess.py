"""The two postprocessors the propagation path itself needs.

``Tautology`` is the default ``personalization_transform`` of every filter (abstract_filters.py:37) and ``Normalize`` is
the callable form of ``use_quotient`` (abstract_filters.py:131-132; the reference's filter tests compare outcomes after
``Normalize``, tests/test_filters.py:41-82).  Behaviour follows pygrank/algorithms/postprocess/postprocess.py:7-80,106-160;
the normalisation constants are device reductions (pgh_reduce / pgh_dot) and the rescale is one elementwise kernel.  The
rest of the postprocessor family re-invokes the hot path."""
from pygrank_amd import backend
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
        if not isinstance(ranker, NodeRanking):
            raise Exception("pygrank can only shift rankers into postprocessors")
        self.ranker = ranker
        return ranker

    # a wrapped filter's collaborators show through, so that ``postprocessor.convergence`` works like the filter's
    preprocessor = property(lambda self: self.ranker.preprocessor)
    convergence = property(lambda self: self.ranker.convergence,
                           lambda self, value: setattr(self.ranker, "convergence", value))


class Tautology(Postprocessor):
    """Changes nothing; without an inner ranker ``rank`` turns its input into a signal."""

    def transform(self, ranks, *args, **kwargs):
        return ranks

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        if self.ranker is None:
            return to_signal(graph, personalization)
        return self.ranker.rank(graph, personalization, *args, **kwargs)


def _is_ranker(obj):
    return callable(getattr(obj, "rank", None))


class Normalize(Postprocessor):
    """Rescales ranks by their maximum ("max", default), sum ("sum"), Euclidean norm ("L2"), or onto [0, 1] ("range").
    ``Normalize("sum", ranker)`` and ``Normalize(ranker, "sum")`` are the same thing (postprocess.py:124-131)."""

    _METHODS = ("max", "sum", "L2", "range")

    def __init__(self, ranker=None, method="max"):
        if ranker is not None and not _is_ranker(ranker):          # the method came first
            ranker, method = (method if _is_ranker(method) else None), ranker
        super().__init__(ranker if ranker is not None else Tautology())
        self.method = method

    def _transform(self, ranks, **kwargs):
        ensure_used_args(kwargs)
        if self.method not in self._METHODS:
            raise Exception("Can only normalize towards max, sum, range, or L2")
        x = ranks.np
        low = float(backend.min(x)) if self.method == "range" else 0.0
        if self.method == "sum":
            high = float(backend.sum(x))
        elif self.method == "L2":
            high = float(backend.dot(x, x)) ** 0.5
        else:
            high = float(backend.max(x))
        if high == low:
            return ranks                                           # constant (or zero) signals stay as they are
        return (x - low) / (high - low)
