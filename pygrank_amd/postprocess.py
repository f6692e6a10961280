"""Minimal postprocessors needed by the propagation path: Tautology (default personalization transform,
abstract_filters.py:37) and Normalize (the callable form of ``use_quotient``, abstract_filters.py:131-132;
the reference's filter tests compare outcomes after ``Normalize``, tests/test_filters.py:41-82).

Restates pygrank/algorithms/postprocess/postprocess.py:7-80,106-160 for these classes only; the rest of
the postprocessor family re-invokes the hot path and is listed as "next" in SURVEY.md 8f.
"""
from pygrank_amd import backend
from pygrank_amd.signals import NodeRanking, to_signal
from pygrank_amd.utils import call, ensure_used_args, remove_used_args


class Postprocessor(NodeRanking):                           # postprocess.py:7-50
    def __init__(self, ranker=None):
        self.ranker = ranker

    def transform(self, ranks, *args, **kwargs):
        return to_signal(ranks, call(self._transform, kwargs, [ranks]))

    def rank(self, *args, **kwargs):
        ranks = self.ranker.rank(*args, **kwargs)
        kwargs = remove_used_args(self.ranker.rank, kwargs)
        return to_signal(ranks, call(self._transform, kwargs, [ranks]))

    def _transform(self, ranks, **kwargs):
        raise Exception("_transform method not implemented for the class " + self.__class__.__name__)

    def _reference(self):
        return self.__class__.__name__

    def references(self):
        if self.ranker is None:
            return [self._reference()]
        refs = self.ranker.references()
        ref = self._reference()
        if ref is not None and len(ref) > 0:
            refs.append(ref)
        return refs

    def __lshift__(self, ranker):
        if not isinstance(ranker, NodeRanking):
            raise Exception("pygrank can only shift rankers into postprocessors")
        self.ranker = ranker
        return ranker

    @property
    def preprocessor(self):
        return self.ranker.preprocessor

    @property
    def convergence(self):
        return self.ranker.convergence

    @convergence.setter
    def convergence(self, value):
        self.ranker.convergence = value


class Tautology(Postprocessor):                              # postprocess.py:53-80
    def __init__(self, ranker=None):
        super().__init__(ranker)

    def transform(self, ranks, *args, **kwargs):
        return ranks

    def rank(self, graph=None, personalization=None, *args, **kwargs):
        if self.ranker is not None:
            return self.ranker.rank(graph, personalization, *args, **kwargs)
        return to_signal(graph, personalization)

    def _reference(self):
        return "tautology" if self.ranker is None else ""


class Normalize(Postprocessor):                              # postprocess.py:106-160
    def __init__(self, ranker=None, method="max"):
        if ranker is not None and not callable(getattr(ranker, "rank", None)):
            ranker, method = method, ranker                  # arguments are re-ordered when swapped
            if not callable(getattr(ranker, "rank", None)):
                ranker = None
        super().__init__(Tautology() if ranker is None else ranker)
        self.method = method

    def _transform(self, ranks, **kwargs):
        ensure_used_args(kwargs)
        x = ranks.np
        low = 0
        if self.method == "range":
            high, low = float(backend.max(x)), float(backend.min(x))
        elif self.method == "max":
            high = float(backend.max(x))
        elif self.method == "sum":
            high = float(backend.sum(x))
        elif self.method == "L2":
            high = float(backend.sum(x ** 2)) ** 0.5
        else:
            raise Exception("Can only normalize towards max, sum, range, or L2")
        if low == high:
            return ranks
        return (x - low) / (high - low)

    def _reference(self):
        if self.method == "range":
            return "[0,1] " + self.method + " normalization"
        return self.method + " normalization"
