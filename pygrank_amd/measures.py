"""Residual measures used as convergence criteria: Mabs (default), L1, MaxDifference.

Restates pygrank/measures/supervised.py:18-47 (Supervised.to_numpy), :93-98 (MaxDifference), :101-106 (Mabs),
:133-138 (L1).  When both operands are HBM vectors the residual is ONE fused HIP reduction
(pgh_residual: |a - b| folded into an f64 sum / max) instead of the reference's three passes
(subtract, abs, sum).  The evaluation measures of the reference (AUC, NDCG, ...) are out of scope
(SURVEY.md 2 rows 18-19).
"""
import ctypes as C
import numbers

from pygrank_amd import _lib as L
from pygrank_amd import backend
from pygrank_amd.device import DeviceVector
from pygrank_amd.signals import GraphSignal, to_signal


class Measure:
    def __call__(self, scores):
        return self.evaluate(scores)

    def evaluate(self, scores):
        raise Exception("Non-abstract subclasses of Measure should implement an evaluate method")


class Supervised(Measure):
    """supervised.py:18-55."""
    _KIND = None

    def __init__(self, known_scores, exclude=None):
        self.known_scores = known_scores
        self.exclude = exclude

    def to_numpy(self, scores, normalization=False):        # supervised.py:36-47
        if isinstance(scores, numbers.Number) and isinstance(self.known_scores, numbers.Number):
            return backend.to_array([self.known_scores]), backend.to_array([scores])
        if isinstance(scores, GraphSignal):
            return to_signal(scores, self.known_scores).filter(exclude=self.exclude), \
                scores.normalized(normalization).filter(exclude=self.exclude)
        if isinstance(self.known_scores, GraphSignal):
            return self.known_scores.filter(exclude=self.exclude), \
                to_signal(self.known_scores, scores).normalized(normalization).filter(exclude=self.exclude)
        if self.exclude is not None:
            raise Exception("Needs to parse graph signal scores or known_scores to be able to exclude specific nodes")
        scores = backend.self_normalize(backend.to_array(scores, copy_array=True)) if normalization \
            else backend.to_array(scores)
        return backend.to_array(self.known_scores), scores

    def evaluate(self, scores):
        known, scores = self.to_numpy(scores)
        if isinstance(known, DeviceVector) and isinstance(scores, DeviceVector):
            out = C.c_double()
            L.check(L.lib().pgh_residual(self._KIND, known._h, scores._h, C.byref(out)))
            return out.value
        raise Exception("residual measures expect backend vectors")


class MaxDifference(Supervised):                             # supervised.py:93-98
    _KIND = L.ERR_LINF


class Mabs(Supervised):                                      # supervised.py:101-106
    _KIND = L.ERR_MABS


class L1(Supervised):                                        # supervised.py:133-138
    _KIND = L.ERR_L1
