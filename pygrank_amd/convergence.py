"""ConvergenceManager: when does the propagation loop stop.

Restates pygrank/algorithms/convergence.py:9-104 -- same constructor, same fields (``iteration``,
``elapsed_time``, ``last_ranks`` kept by reference), same stopping rule.  The fused device loops
(pygrank_amd/filters.py -> pgh_ppr_run / pgh_poly_run / pgh_absorb_run) evaluate the identical rule on the
GPU and write ``iteration`` / ``elapsed_time`` back into this object, so callers that inspect
``ranker.convergence`` (tests/test_filters.py:33-38) see the reference's bookkeeping.
"""
from timeit import default_timer as time

from pygrank_amd import _lib as L
from pygrank_amd import backend
from pygrank_amd.measures import L1, Mabs, MaxDifference


class ConvergenceManager:
    """convergence.py:9-104.  ``has_converged`` is asked once BEFORE every step; it counts the call, gives up at
    ``max_iters`` (silently for ``error_type="iters"`` or without an exception type, loudly otherwise) and otherwise
    compares the iterate with the one it saw last time."""

    def __init__(self, tol=1.E-6, error_type=Mabs, max_iters=100, end_modulo=1, iter_exception=Exception):
        self.tol, self.error_type = tol, error_type
        self.max_iters, self.end_modulo = max_iters, end_modulo
        self.iter_exception = iter_exception
        self.iteration, self.last_ranks = 0, None
        self.elapsed_time = self._start_time = None

    def start(self, restart_timer=True):
        self.last_ranks = None
        if self._start_time is not None and not restart_timer:
            return                                          # a nested run keeps the clock and the count (convergence.py:62-75)
        self._start_time, self.elapsed_time, self.iteration = time(), None, 0

    def _out_of_iterations(self):
        return "Could not converge within " + str(self.max_iters) + " iterations"

    def _counts_only(self):
        return self.error_type == "iters"

    def has_converged(self, new_ranks):
        self.iteration += 1
        stop = False
        if self.iteration >= self.max_iters:
            if not self._counts_only() and self.iter_exception is not None:
                raise self.iter_exception(self._out_of_iterations())
            stop = True
        else:
            previous, self.last_ranks = self.last_ranks, new_ranks
            stop = previous is not None and self._has_converged(previous, new_ranks)
        self._stamp()
        return stop

    def _stamp(self):
        """elapsed_time = seconds since start(): refreshed by every has_converged call and at the exit of a device loop."""
        self.elapsed_time = time() - self._start_time

    def _has_converged(self, prev_ranks, ranks):
        """The comparison itself: skipped on iterations that are not a multiple of end_modulo and in counting mode."""
        if self._counts_only() or self.iteration % self.end_modulo:
            return False
        return self.error_type(prev_ranks)(ranks) <= self.effective_tolerance()

    def effective_tolerance(self):                           # convergence.py:101: never below the engine's epsilon
        return 0 if self.tol is None else max(self.tol, backend.epsilon())

    # ---- device-loop plumbing ----------------------------------------------------------------------
    def device_error_kind(self):
        """PGH_ERR_* code when the stopping rule can be evaluated on the device, else None."""
        if self._counts_only():
            return L.ERR_ITERS
        for cls, kind in ((Mabs, L.ERR_MABS), (L1, L.ERR_L1), (MaxDifference, L.ERR_LINF)):
            if self.error_type is cls:
                return kind
        return None

    def finish_device_loop(self, iterations, converged):
        """What has_converged leaves behind at loop exit, for a loop that ran on the device."""
        self.iteration = int(iterations)
        self._stamp()
        ran_out = not converged and self.iteration >= self.max_iters
        if ran_out and not self._counts_only() and self.iter_exception is not None:
            raise self.iter_exception(self._out_of_iterations())

    def __str__(self):
        return f"{self.iteration} iterations ({self.elapsed_time} sec)"
