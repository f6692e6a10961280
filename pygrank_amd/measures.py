"""Residual measures used as convergence criteria -- Mabs (default), L1, MaxDifference -- and the residual-style supervised
measures on vectors already in HBM (RMabs, MSQ, MSQRT, L2, Euclidean, Cos, Dot; supervised.py:109-154,208-222).

Restates pygrank/measures/supervised.py:18-47 (Supervised.to_numpy), :93-98 (MaxDifference), :101-106 (Mabs),
:133-138 (L1).  When both operands are HBM vectors the residual is ONE fused HIP reduction
(pgh_residual: |a - b| folded into an f64 sum / max) instead of the reference's three passes
(subtract, abs, sum).  AUC (supervised.py:255-263) is one device sort (pgh_auc); the other evaluation measures of the
reference (NDCG, ...) are out of scope.
"""
import ctypes as C
import numbers

from pygrank_amd import _lib as L
from pygrank_amd import backend
from pygrank_amd.device import DeviceVector, lazy_residual
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

    def to_numpy(self, scores, normalization=False):
        """supervised.py:36-47 -> (known, scores) as two aligned vectors.  Two plain numbers become one-element vectors.  When
        either side is a graph signal it lends its graph to the other, the scores are (optionally) normalised and both lose the
        excluded nodes; two plain vectors are taken as they are (nothing to exclude nodes by)."""
        known = self.known_scores
        if all(isinstance(value, numbers.Number) for value in (scores, known)):
            return backend.to_array([known]), backend.to_array([scores])
        anchor = next((value for value in (scores, known) if isinstance(value, GraphSignal)), None)
        if anchor is not None:
            aligned = (to_signal(anchor, known), to_signal(anchor, scores).normalized(normalization))
            return tuple(signal.filter(exclude=self.exclude) for signal in aligned)
        if self.exclude is not None:
            raise Exception("Needs to parse graph signal scores or known_scores to be able to exclude specific nodes")
        plain = backend.to_array(scores, copy_array=bool(normalization))
        return backend.to_array(known), (backend.self_normalize(plain) if normalization else plain)

    def evaluate(self, scores):
        known, scores = self.to_numpy(scores)
        if isinstance(known, DeviceVector) and isinstance(scores, DeviceVector):
            # two iterates of one graph that are still in the engine's id space (device.LazyVector): the residual is taken there
            resident = lazy_residual(self._KIND, known, scores)
            if resident is not None:
                return resident
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


class _Pairwise(Supervised):
    """Measures that are a handful of device reductions over the two score vectors."""

    def _pair(self, scores):
        known, scores = self.to_numpy(scores)
        if not (isinstance(known, DeviceVector) and isinstance(scores, DeviceVector)):
            raise Exception("residual measures expect backend vectors")
        return known, scores

    def _squared_distance(self, scores):
        known, scores = self._pair(scores)
        d = known - scores
        return d.dot(d), len(scores)


class RMabs(_Pairwise):                                      # supervised.py:109-114
    def evaluate(self, scores):
        known, scores = self._pair(scores)
        out = C.c_double()
        L.check(L.lib().pgh_residual(L.ERR_L1, known._h, scores._h, C.byref(out)))
        return out.value / known.abssum()


class MSQ(_Pairwise):                                        # supervised.py:117-122
    def evaluate(self, scores):
        total, n = self._squared_distance(scores)
        return total / n


class MSQRT(_Pairwise):                                      # supervised.py:125-130
    def evaluate(self, scores):
        total, n = self._squared_distance(scores)
        return (total / n) ** 0.5


class L2(_Pairwise):                                         # supervised.py:141-146 (the squared distance, as the reference)
    def evaluate(self, scores):
        return self._squared_distance(scores)[0]


class Euclidean(_Pairwise):                                  # supervised.py:149-154
    def evaluate(self, scores):
        return self._squared_distance(scores)[0] ** 0.5


class Cos(_Pairwise):                                        # supervised.py:208-214
    def evaluate(self, scores):
        known, scores = self._pair(scores)
        return backend.safe_div(known.dot(scores), (known.dot(known) * scores.dot(scores)) ** 0.5)


class AUC(_Pairwise):                                        # supervised.py:255-263 (sklearn roc_curve + auc in the reference)
    """Area under the ROC curve of the scores against binary known scores, ties at their mid-rank, with ONE device sort
    (pgh_auc) instead of a trip through sklearn on the host."""

    def evaluate(self, scores):
        known, scores = self._pair(scores)
        out, positives = C.c_double(), C.c_int64()
        L.check(L.lib().pgh_auc(known._h, scores._h, C.byref(out), C.byref(positives)))
        if positives.value == 0 or positives.value == len(scores):
            raise Exception("Cannot evaluate AUC when all labels are the same")
        return out.value


class Dot(_Pairwise):                                        # supervised.py:217-222
    def evaluate(self, scores):
        known, scores = self._pair(scores)
        return known.dot(scores)
