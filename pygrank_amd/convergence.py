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
    def __init__(self, tol=1.E-6, error_type=Mabs, max_iters=100, end_modulo=1, iter_exception=Exception):
        self.tol = tol
        self.error_type = error_type
        self.max_iters = max_iters
        self.iteration = 0
        self.last_ranks = None
        self._start_time = None
        self.elapsed_time = None
        self.iter_exception = iter_exception
        self.end_modulo = end_modulo

    def start(self, restart_timer=True):                     # convergence.py:62-75
        if restart_timer or self._start_time is None:
            self._start_time = time()
            self.elapsed_time = None
            self.iteration = 0
        self.last_ranks = None

    def has_converged(self, new_ranks):                      # convergence.py:77-94
        self.iteration += 1
        if self.iteration >= self.max_iters:
            if self.error_type == "iters" or self.iter_exception is None:
                self.elapsed_time = time() - self._start_time
                return True
            raise self.iter_exception("Could not converge within " + str(self.max_iters) + " iterations")
        converged = False if self.last_ranks is None else self._has_converged(self.last_ranks, new_ranks)
        self.last_ranks = new_ranks
        self.elapsed_time = time() - self._start_time
        return converged

    def _has_converged(self, prev_ranks, ranks):             # convergence.py:96-101
        if self.error_type == "iters":
            return False
        if self.iteration % self.end_modulo != 0:
            return False
        return self.error_type(prev_ranks)(ranks) <= self.effective_tolerance()

    def effective_tolerance(self):                           # convergence.py:101
        return 0 if self.tol is None else max(self.tol, backend.epsilon())

    # ---- device-loop plumbing ----------------------------------------------------------------------
    def device_error_kind(self):
        """PGH_ERR_* code when the stopping rule can be evaluated on the device, else None."""
        if self.error_type == "iters":
            return L.ERR_ITERS
        for cls, kind in ((Mabs, L.ERR_MABS), (L1, L.ERR_L1), (MaxDifference, L.ERR_LINF)):
            if self.error_type is cls:
                return kind
        return None

    def finish_device_loop(self, iterations, converged):
        """Mirror of the loop exit of has_converged for a loop that ran on the device."""
        self.iteration = int(iterations)
        self.elapsed_time = time() - self._start_time
        if not converged and self.error_type != "iters" and self.iter_exception is not None \
                and self.iteration >= self.max_iters:
            raise self.iter_exception("Could not converge within " + str(self.max_iters) + " iterations")

    def __str__(self):
        return str(self.iteration) + " iterations (" + str(self.elapsed_time) + " sec)"
