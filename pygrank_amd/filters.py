"""Graph filters of the propagation path on the MI355X engine.

Same classes, constructor arguments, operators and error behaviour as the reference:
  GraphFilter / RecursiveGraphFilter / ClosedFormGraphFilter  pygrank/algorithms/filters/abstract_filters.py:11-270
  PageRank, PageRankClosed, HeatKernel, AbsorbingWalks         pygrank/algorithms/filters/adhoc.py:9-174
  GenericGraphFilter, ImpulseGraphFilter, LowPassRecursive...  pygrank/algorithms/filters/low_pass.py:5-91

Two execution routes, both on the GPU:
  * generic route -- the reference's ``_start / while not has_converged: _step / _end`` structure
    (abstract_filters.py:58-62) expressed with backend primitives; used whenever an option needs per-step
    Python (callable quotient, custom error measure, optimisation dict, graph dropout ...).
  * fused route -- the whole loop runs inside the engine (pgh_ppr_run / pgh_absorb_run / pgh_poly_run,
    include/pgh.h): one merge-path SpMV kernel per iteration with the filter's vector algebra in its
    epilogue, the residual as a separate wavefront-reduced kernel, and ConvergenceManager's stopping rule
    evaluated on the device so that the host does not synchronise every iteration.  Results and
    ``convergence.iteration`` are those of the generic route.
"""
import ctypes as C

import numpy as np

from pygrank_amd import _lib as L
from pygrank_amd import backend
from pygrank_amd.convergence import ConvergenceManager
from pygrank_amd.device import DeviceGraph, DeviceVector, LazyVector
from pygrank_amd.postprocess import Postprocessor, Tautology
from pygrank_amd.preprocessing import obj2id, preprocessor as default_preprocessor
from pygrank_amd.signals import GraphSignal, NodeRanking, to_signal
from pygrank_amd.utils import call, ensure_used_args


def _device_graph(M):
    array = getattr(M, "array", M)
    return array if isinstance(array, DeviceGraph) else None


def _f64_image_usable(g):
    """The engine's f64 image serves square graphs that carry the blocked layout (csrc/pgh_bsf64.hip bsf64_usable)."""
    return g is not None and g.shape[0] == g.shape[1] and g.shape[0] > 0 and g.nnz > 0 and "row-major" not in g.format()


class GraphFilter(NodeRanking):
    """abstract_filters.py:11-106."""

    # collaborators a filter builds for itself from its keyword arguments when none is handed in: attribute -> factory
    _BUILT_FROM_KWARGS = (("preprocessor", default_preprocessor), ("convergence", ConvergenceManager))

    def __init__(self, preprocessor=None, convergence=None, personalization_transform=None, preserve_norm=True,
                 dtype=None, **kwargs):
        """dtype (an extension of this backend; the reference's numpy engine is fp64 throughout, pygrank/core/backend/numpy.py:84-86):
        None -- the engine's f32 loop, and f64 iterates / sums / quotient / residual on the engine's f64 image whenever the tolerance lies
        below fp32 eps (where an f32 loop, clamped at fp32 eps by convergence.py:101, stops early: the reference's AbsorbingWalks default
        alpha = 1 - 1e-6 with tol = 1e-9 takes 21 iterations, an f32 loop 12); "float32" -- always the f32 loop; "float64" -- always f64."""
        if dtype not in (None, "float32", "float64"):
            raise Exception("dtype is None (f32, f64 below fp32 eps), 'float32' or 'float64'")
        self.dtype = dtype
        given = dict(preprocessor=preprocessor, convergence=convergence)
        for attribute, factory in self._BUILT_FROM_KWARGS:
            setattr(self, attribute, given[attribute] if given[attribute] is not None else call(factory, kwargs))
        ensure_used_args(kwargs, [factory for _, factory in self._BUILT_FROM_KWARGS])   # a keyword nobody declares is a typo
        self.personalization_transform = personalization_transform or Tautology()
        self.preserve_norm = preserve_norm

    def _prepare(self, personalization):
        pass

    def rank(self, graph=None, personalization=None, warm_start=None, graph_dropout=0, *args, **kwargs):
        """abstract_filters.py:44-65.  personalization -> signal -> (transform) -> L1-normalised seed vector; the iterate starts
        as a copy of it (or of warm_start); the preprocessor turns the graph into the operator; the loop runs until the
        convergence manager says stop; preserve_norm gives the input's L1 norm back.  Whenever the configuration allows it the
        whole loop is ONE engine call (_fused_rank / _fused_loop); the hook protocol below serves everything else."""
        seed_signal = to_signal(graph, personalization)
        self._prepare(seed_signal)
        signal = self.personalization_transform(seed_signal)
        if warm_start is None:
            # norm, normalisation and start vector folded into the engine's first pass over the operands (pgh_loop_cfg.in_norm < 0,
            # start_from_p): no reduction and no host round trip in front of the first step.  A zero norm comes back as the signal.
            # graph_dropout > 0: the same loop with a fresh mask per step inside the step's kernels (pgh_ppr_run_dropout).
            fused = self._fused_rank(signal, *args, **kwargs) if graph_dropout == 0 else \
                self._fused_rank(signal, *args, dropout=graph_dropout, **kwargs)
            if fused is not None:
                return fused
        raw = signal.np
        norm = raw.abssum() if isinstance(raw, DeviceVector) else backend.sum(backend.abs(raw))
        if norm == 0:
            return signal                       # nothing to spread: the reference hands the (all-zero) signal back
        # (the f64 routes of the walks normalise the personalization themselves, in f64: an f32 quotient p / |p| sums to 1 only within
        # 6e-8, which a run at tol = 1e-9 sees -- it would start beside the fixed point the reference starts ON)
        self._raw_input = (raw, float(norm)) if warm_start is None and isinstance(raw, DeviceVector) else None
        signal = to_signal(signal, signal.np / norm)
        # the reference never writes into the caller's warm_start (every step builds a fresh array); the device loops update
        # their iterate in place, so the start vector is always a copy
        ranks = to_signal(signal, backend.copy(signal.np if warm_start is None else to_signal(signal, warm_start).np))
        M = self.preprocessor(self._prepare_graph(signal.graph, signal, *args, **kwargs))
        self.convergence.start()
        try:
            fused = graph_dropout == 0 and self._fused_loop(M, signal, ranks, norm if self.preserve_norm else 1.0, *args, **kwargs)
        finally:
            self._raw_input = None             # (only the f64 routes of _fused_loop look at it; nothing keeps the caller's vector alive)
        if fused:
            return ranks
        self._host_driven_loop(M, signal, ranks, graph_dropout, args, kwargs)
        if self.preserve_norm:
            ranks.np = ranks.np * norm
        return ranks

    def _host_driven_loop(self, M, signal, ranks, dropout_rate, args, kwargs):
        """The hook protocol of abstract_filters.py:57-62 for filters and configurations without a device loop: _start, then
        _step until the convergence manager stops the loop, then _end -- each hook sees an operator whose entries were dropped
        afresh (backend.graph_dropout draws a new mask per call; rate 0 returns the operator itself)."""
        def operator():
            return backend.graph_dropout(M, dropout_rate)
        self._start(operator(), signal, ranks, *args, **kwargs)
        keep_going = lambda: not self.convergence.has_converged(ranks.np)      # noqa: E731
        while keep_going():
            self._step(operator(), signal, ranks, *args, **kwargs)
        self._end(operator(), signal, ranks, *args, **kwargs)

    # ---- hooks
    def _fused_rank(self, personalization, *args, dropout=0, **kwargs):
        """Whole rank() inside the engine, norm included, starting from the UN-normalised personalization; returns the ranks
        signal (the personalization itself when its norm is zero) or None when this filter / configuration has no such route.
        dropout: graph_dropout of every step."""
        return None

    def _fused_loop(self, M, personalization, ranks, out_scale, *args, **kwargs):
        """Runs the whole loop inside the engine when the configuration allows it; returns True when it did."""
        return False

    def _f64_wanted(self):
        """f64 iterates for this run (see __init__): asked for, or chosen because the tolerance lies below fp32 eps."""
        if self.dtype == "float64":
            return True
        cm = self.convergence
        if self.dtype == "float32" or type(cm) is not ConvergenceManager or cm.device_error_kind() in (None, L.ERR_ITERS):
            return False
        return cm.tol is None or float(cm.tol) < backend.epsilon()

    def _f64_cfg(self, cfg):
        """the loop configuration of an f64 route: the tolerance never below fp64 eps (convergence.py:101 with numpy's epsilon())"""
        tol = self.convergence.tol
        cfg.tol = 0.0 if tol is None else max(float(tol), float(np.finfo(np.float64).eps))
        return cfg

    def _f64_operands(self, cfg, p):
        """(cfg, personalization) of an f64 recursive route: the UN-normalised personalization with its norm beside it whenever rank()
        kept them (no warm start) -- the engine then forms p / |p| and the start vector in f64 (pgh_loop_cfg.in_norm, start_from_p)"""
        cfg = self._f64_cfg(cfg)
        raw = getattr(self, "_raw_input", None)
        if raw is not None and len(raw[0]) == len(p) and raw[1] > 0:
            cfg.in_norm, cfg.start_from_p = raw[1], 1
            return cfg, raw[0]
        return cfg, p

    def _loop_cfg(self, alpha=0.0, use_quotient=False, out_scale=1.0):
        cm = self.convergence
        if type(cm) is not ConvergenceManager:
            return None
        kind = cm.device_error_kind()
        if kind is None:
            return None
        return L.LoopCfg(alpha=float(alpha), use_quotient=1 if use_quotient else 0, err_kind=kind,
                         tol=float(cm.effective_tolerance()), max_iters=int(cm.max_iters),
                         end_modulo=int(cm.end_modulo), out_scale=float(out_scale))

    def _prepare_graph(self, graph, *args, **kwargs):
        return graph

    def _start(self, M, personalization, ranks, *args, **kwargs):
        pass

    def _end(self, M, personalization, ranks, *args, **kwargs):
        pass

    def _step(self, M, personalization, ranks, *args, **kwargs):
        raise Exception("Use a derived class of GraphFilter that implements the _step method")

    def _slot_for(self, part):
        """Which attribute `filter + part` replaces (abstract_filters.py:86-95): a convergence manager, a preprocessor
        (recognised by the name preprocessor() gives its outcome) or a postprocessor that takes the quotient's place."""
        if isinstance(part, ConvergenceManager):
            return "convergence"
        if getattr(part, "__name__", None) == "preprocess":
            return "preprocessor"
        if isinstance(part, Postprocessor):
            return "use_quotient"
        return None

    def __add__(self, part):
        slot = self._slot_for(part)
        if slot is None:
            raise Exception("a graph filter takes convergence managers, preprocessors and quotient postprocessors, not "
                            + type(part).__name__)
        setattr(self, slot, part)
        return self

    def __lshift__(self, ranker):
        """`ranker >> filter`: the ranker's outcome becomes this filter's personalization (abstract_filters.py:97-101)."""
        if isinstance(ranker, NodeRanking):
            self.personalization_transform = ranker
            return ranker
        raise Exception("only node ranking algorithms can be chained into a graph filter")

    def _reference(self):
        return "graph filter"

    def cite(self):                                                               # abstract_filters.py:82-86
        own = super().cite()
        upstream = self.personalization_transform
        if isinstance(upstream, Tautology) and upstream.ranker is None:
            return own
        return upstream.cite() + "\n  passed to " + own


# =====================================================================================================
# recursive filters
# =====================================================================================================
class RecursiveGraphFilter(GraphFilter):
    """abstract_filters.py:104-149: ranks = formula(G, ranks), optional L1 quotient per step."""

    def __init__(self, use_quotient=True, converge_to_eigenvectors=False, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.use_quotient, self.converge_to_eigenvectors = use_quotient, converge_to_eigenvectors

    def _quotient(self, ranks):
        """What follows the formula in every step (abstract_filters.py:130-134): a postprocessor in the quotient's place,
        the plain division by the sum, or nothing."""
        rule = self.use_quotient
        if isinstance(rule, Postprocessor):
            return rule(ranks)
        return backend.safe_div(ranks, backend.sum(ranks)) if rule else ranks.np

    def _step(self, M, personalization, ranks, *args, **kwargs):
        outcome = self._formula(M, personalization, ranks, *args, **kwargs)
        ranks.np = outcome.np if isinstance(outcome, GraphSignal) else outcome
        ranks.np = self._quotient(ranks)
        if self.converge_to_eigenvectors:                   # the personalization follows the iterate (abstract_filters.py:135-136)
            personalization.np = ranks.np

    def references(self):
        extra = ["unbiased convergence to the eigenvector"] if self.converge_to_eigenvectors else []
        return super().references() + extra

    def _formula(self, M, personalization, ranks, *args, **kwargs):
        raise Exception("Use a derived class of RecursiveGraphFilter that implements the _formula method")

    def _plain_quotient(self):
        return (self.use_quotient is None or isinstance(self.use_quotient, (bool, int))) \
            and not self.converge_to_eigenvectors

    def _run_recursive(self, entry, g, cfg, ranks, *vectors):
        if cfg is None or g is None or g.shape[0] != g.shape[1]:
            return False
        x = ranks.np
        if not isinstance(x, DeviceVector):
            return False
        if isinstance(x, LazyVector):
            x = x._writable()              # the engine updates the iterate in place: an expression becomes memory of its own first
        x._before_write()
        res = L.LoopResult()
        L.check(entry(g._h, *[v._h for v in vectors], x._h, C.byref(cfg), C.byref(res)))
        if cfg.in_norm < 0 and res.in_norm == 0:
            return "zero"                  # an all-zero personalization: nothing was run, nothing is counted
        ranks.np = x                       # updated in place by the engine; drops the host mirror
        self.last_loop = dict(iterations=res.iterations, converged=bool(res.converged), spmv=res.spmv_count,
                              last_error=res.last_error, loop_ms=res.loop_ms, flags=res.flags)
        self.convergence.finish_device_loop(res.iterations, res.converged)
        return True


class PageRank(RecursiveGraphFilter):
    """adhoc.py:9-46: r <- alpha * M^T r + (1 - alpha) * p."""

    def __init__(self, alpha=0.85, *args, dtype=None, **kwargs):
        """dtype="float64" (an extension of this backend): the loop keeps its iterates, sums, quotient and residual in f64 on the engine's
        f64 image (pgh_ppr_run_f64) and the tolerance is clamped at fp64 eps like the reference's numpy backend
        (pygrank/core/backend/numpy.py:84-86) instead of fp32 eps -- e.g. tol=1e-9 as in the reference's tests/test_filters.py:189,194.
        An exactness mode (one host look per step); personalization and ranks stay f32 vectors."""
        self.alpha = alpha
        super().__init__(*args, dtype=dtype, **kwargs)

    def _reference(self):
        return f"personalized PageRank (restart probability {1 - self.alpha:.3g})"

    def _formula(self, M, personalization, ranks, *args, **kwargs):               # adhoc.py:34-36
        return backend.conv(ranks, M) * self.alpha + personalization * (1 - self.alpha)

    def _fused_loop(self, M, personalization, ranks, out_scale, *args, **kwargs):
        if args or kwargs or not self._plain_quotient() or type(self)._formula is not PageRank._formula \
                or type(self)._step is not RecursiveGraphFilter._step:
            return False
        cfg = self._loop_cfg(self.alpha, bool(self.use_quotient), out_scale)
        p = personalization.np
        if not isinstance(p, DeviceVector):
            return False
        if self.dtype == "float64" and cfg is None:
            raise Exception("PageRank(dtype='float64') needs a stopping rule the engine evaluates (Mabs / L1 / MaxDifference / 'iters')")
        g = _device_graph(M)
        if cfg is not None and self._f64_wanted() and _f64_image_usable(g):
            return self._run_recursive(L.lib().pgh_ppr_run_f64, g, self._f64_cfg(cfg), ranks, p)
        return self._run_recursive(L.lib().pgh_ppr_run, g, cfg, ranks, p)

    fused_dropout = True          # False (on an instance): rank(..., graph_dropout=) takes the hook protocol, one engine call per primitive

    def _fused_rank(self, personalization, *args, dropout=0, **kwargs):
        if args or kwargs or not self._plain_quotient() or type(self)._formula is not PageRank._formula \
                or type(self)._step is not RecursiveGraphFilter._step or type(self)._prepare_graph is not GraphFilter._prepare_graph:
            return None
        if dropout and not (self.fused_dropout and 0 < dropout < 1):
            return None
        if self.dtype == "float64" and dropout:
            raise Exception("PageRank(dtype='float64') has no graph_dropout route")
        p = personalization.np
        cfg = self._loop_cfg(self.alpha, bool(self.use_quotient), -1.0 if self.preserve_norm else 1.0)     # -1: times the norm
        if cfg is None or not isinstance(p, DeviceVector):
            return None
        M = self.preprocessor(personalization.graph)
        g = _device_graph(M)
        if g is None or g.shape[0] != g.shape[1] or g.shape[0] != len(p):
            return None
        cfg.in_norm, cfg.start_from_p = -1.0, 1            # the engine sums |p| itself
        before = dict(vars(self.convergence))
        self.convergence.start()
        ranks = to_signal(personalization, DeviceVector.empty(len(p)))
        entry = L.lib().pgh_ppr_run
        if self._f64_wanted() and not dropout and _f64_image_usable(g):
            entry, cfg = L.lib().pgh_ppr_run_f64, self._f64_cfg(cfg)
        if dropout:
            # the masks the hook protocol would draw (abstract_filters.py:57-62): one for _start, one per step, one for _end -- step k
            # runs on the (k + 1)-th of them, so both routes compute the same thing from the same seed
            from pygrank_amd.backend import hip as _hip
            first = _hip.peek_dropout_seed()
            lib, rate, seed0 = L.lib(), float(dropout), int(first) + 1

            def entry(gh, ph, xh, cfg_ref, res_ref):
                return lib.pgh_ppr_run_dropout(gh, ph, xh, cfg_ref, rate, seed0, res_ref)
        outcome = self._run_recursive(entry, g, cfg, ranks, p)
        if outcome == "zero":                              # abstract_filters.py:53-54: returned before the manager is started
            vars(self.convergence).update(before)
            return personalization
        if dropout and outcome:
            # ... and leaves the seed counter where the hook protocol would: it draws iterations + 1 masks (_start, iterations - 1
            # steps, _end), however early the tolerance stopped the run; a run that fell back (outcome False) has drawn none
            _hip.take_dropout_seeds(int(self.convergence.iteration) + 1)
        return ranks if outcome else None

    def propagate(self, graph, features, *args, **kwargs):
        """signals.py:225-226 semantics (one rank() per feature column), run as multi-seed batches of up to 64 columns
        through pgh_ppr_run_batch: the adjacency is streamed once per iteration for the whole batch and every column
        keeps its own quotient, residual and stopping iteration."""
        from pygrank_amd.device import DeviceMatrix
        cfg = self._loop_cfg(self.alpha, bool(self.use_quotient), 1.0)
        dropout = float(kwargs.get("graph_dropout", 0) or 0)
        only_dropout = not kwargs or (set(kwargs) == {"graph_dropout"} and 0 <= dropout < 1)
        batched = (not args and only_dropout and cfg is not None and self._plain_quotient()
                   and type(self)._formula is PageRank._formula and type(self)._step is RecursiveGraphFilter._step
                   and isinstance(self.personalization_transform, Tautology) and self.personalization_transform.ranker is None)
        if self.dtype == "float64":
            #  the multi-seed loop is an f32 loop: a ranker that promises f64 iterates and an fp64-eps tolerance runs its columns
            # one by one through pgh_ppr_run_f64 (signals.py:225-226 as written) instead of silently changing precision
            batched = False
        M = self.preprocessor(graph) if batched else None
        g = _device_graph(M) if batched else None
        if g is None or g.shape[0] != g.shape[1]:
            return super().propagate(graph, features, *args, **kwargs)
        F = features if isinstance(features, DeviceMatrix) else backend.to_primitive(features)
        if not isinstance(F, DeviceMatrix):
            F = DeviceMatrix.from_columns([F])
        self.last_batches = []
        out = DeviceMatrix.empty(F.n, F.b) if F.b > 64 else None
        for start in range(0, F.b, 64):
            chunk = F if F.b <= 64 else F.get_cols(start, min(64, F.b - start))
            norms = chunk.col_abssum()                                            # abstract_filters.py:52-55 per column
            P = chunk.div_cols(norms)                                             # zero columns stay zero (:53-54)
            R = DeviceMatrix.empty(chunk.n, chunk.b)
            results = (L.LoopResult * chunk.b)()
            scales = (C.c_double * chunk.b)(*[(float(nrm) if self.preserve_norm else 1.0) for nrm in norms])
            cfg.start_from_p = 1                                                  # ranks start as a copy of p (:56)
            self.convergence.start()
            if dropout > 0:
                # graph_dropout(M, rate) of every step (abstract_filters.py:59-62) inside the batch kernel: one mask per step,
                # seeds reserved up front (the loop stops on the device)
                from pygrank_amd.backend import hip as _hip
                seed0 = _hip.take_dropout_seeds(max(int(cfg.max_iters), 1))
                L.check(L.lib().pgh_ppr_run_batch_dropout(g._h, P._h, R._h, C.byref(cfg), scales, dropout, seed0, results))
            else:
                L.check(L.lib().pgh_ppr_run_batch(g._h, P._h, R._h, C.byref(cfg), scales, results))
            info = [dict(iterations=r.iterations, converged=bool(r.converged), spmv=r.spmv_count, loop_ms=r.loop_ms)
                    for r in results]
            self.last_batches.append(info)
            for r, nrm in zip(results, norms):
                if nrm != 0:
                    self.convergence.finish_device_loop(r.iterations, r.converged)   # raises like the per-column run would
            if out is None:
                return R
            out.set_cols(start, R)
        return out


class AbsorbingWalks(RecursiveGraphFilter):
    """adhoc.py:125-174: partially absorbing random walks."""

    def __init__(self, alpha=1 - 1.E-6, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha = alpha

    def _reference(self):
        return f"partially absorbing random walks (absorption scale {(1 - self.alpha) / self.alpha:.3g})"

    def _start(self, M, personalization, ranks, absorption=None, **kwargs):       # adhoc.py:157-159
        self.absorption = to_signal(personalization.graph, absorption) * ((1 - self.alpha) / self.alpha)
        self.degrees = backend.degrees(M)

    def _end(self, *args, **kwargs):
        super()._end(*args, **kwargs)
        del self.absorption
        del self.degrees

    def _formula(self, M, personalization, ranks, *args, **kwargs):               # adhoc.py:166-169
        return (backend.conv(ranks, M) * self.degrees + personalization * self.absorption) / \
            (self.absorption + self.degrees)

    def _fused_loop(self, M, personalization, ranks, out_scale, *args, absorption=None, **kwargs):
        if args or kwargs or not self._plain_quotient() or type(self)._formula is not AbsorbingWalks._formula \
                or type(self)._step is not RecursiveGraphFilter._step:
            return False
        cfg = self._loop_cfg(self.alpha, bool(self.use_quotient), out_scale)
        p = personalization.np
        if not isinstance(p, DeviceVector):
            return False
        lam = (to_signal(personalization.graph, absorption) * ((1 - self.alpha) / self.alpha)).np
        g = _device_graph(M)
        if cfg is not None and self._f64_wanted() and _f64_image_usable(g):
            cfg, p = self._f64_operands(cfg, p)
            return self._run_recursive(L.lib().pgh_absorb_run_f64, g, cfg, ranks, p, lam)
        return self._run_recursive(L.lib().pgh_absorb_run, g, cfg, ranks, p, lam)


class SymmetricAbsorbingRandomWalks(RecursiveGraphFilter):
    """adhoc.py:317-369: symmetric partially absorbing random walks.  With d = degrees(M) and the per-node absorption
    a = (1 + sqrt(1 + 4 d)) / 2 a step is ``conv(ranks / a, M) * d / (a + d) + personalization * a / (a + d)``
    (``alpha`` is accepted for signature compatibility and, as in the reference, does not enter the formula)."""

    def __init__(self, alpha=0.5, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.alpha = alpha

    def _reference(self):
        return "symmetric partially absorbing random walks"

    def _start(self, M, personalization, ranks, **kwargs):                        # adhoc.py:348-353
        d = backend.degrees(M)
        a = ((d * 4 + 1) ** 0.5 + 1) / 2
        self._walk = dict(pre=1. / a, post=d / (a + d), skew=a / (a + d))

    def _end(self, *args, **kwargs):
        super()._end(*args, **kwargs)
        self._walk = None

    def _formula(self, M, personalization, ranks, *args, **kwargs):               # adhoc.py:362-364
        w = self._walk
        return backend.conv(ranks * w["pre"], M) * w["post"] + personalization * w["skew"]

    def _fused_loop(self, M, personalization, ranks, out_scale, *args, **kwargs):
        if args or kwargs or not self._plain_quotient() or type(self)._formula is not SymmetricAbsorbingRandomWalks._formula \
                or type(self)._step is not RecursiveGraphFilter._step or type(self)._start is not SymmetricAbsorbingRandomWalks._start:
            return False
        cfg = self._loop_cfg(self.alpha, bool(self.use_quotient), out_scale)
        p = personalization.np
        g = _device_graph(M)
        if not isinstance(p, DeviceVector) or g is None or "row-major" in g.format():     # PGH_FORMAT=csr: generic route
            return False
        if cfg is not None and self._f64_wanted() and _f64_image_usable(g):
            cfg, p = self._f64_operands(cfg, p)
            return self._run_recursive(L.lib().pgh_sarw_run_f64, g, cfg, ranks, p)
        return self._run_recursive(L.lib().pgh_sarw_run, g, cfg, ranks, p)


class LowPassRecursiveGraphFilter(GraphFilter):
    """low_pass.py:61-91: recursive low-pass filter with per-iteration parameters."""

    def __init__(self, params=None, *args, **kwargs):
        if params is None:
            params = [0.9] * 10
        super().__init__(*args, **kwargs)
        self.params = params

    def _step(self, M, personalization, ranks, *args, **kwargs):
        if self.convergence.iteration > len(self.params):
            return 0
        param = self.params[self.convergence.iteration - 1]
        if param == 0:
            return ranks
        if param == 1:
            ranks.np = backend.conv(ranks, M).np
            return ranks
        ranks.np = (backend.conv(ranks, M) * param + personalization * (1 - param)).np


class ImpulseGraphFilter(GraphFilter):
    """low_pass.py:29-58: filter defined by its impulse-response parameters."""

    def __init__(self, params=None, *args, **kwargs):
        if params is None:
            params = [0.9] * 10
        super().__init__(*args, **kwargs)
        self.params = params

    def _step(self, M, personalization, ranks, *args, **kwargs):
        if self.convergence.iteration > len(self.params):
            return 0
        param = self.params[self.convergence.iteration - 1]
        if param == 0:
            return ranks
        if param == 1:
            ranks.np = backend.conv(ranks, M).np
            return ranks
        ranks.np = (backend.conv(ranks, M) * param + ranks * (1 - param)).np


# =====================================================================================================
# closed-form (polynomial) filters
# =====================================================================================================
class _PowerSlab:
    """The device form of the optimisation dict (abstract_filters.py:232-246): the powers
    {(M^T)^k p, k = 0, 1, ...} of ONE personalization as the columns of [n, 64] slabs in HBM (as many slabs as the
    longest expansion asked for needs), extended on demand (one conv per new column), with the L1 and L-inf norm of every
    column on the host.  A filter evaluated from it costs one pass over the slabs (pgh_mat_gemv) instead of one SpMV
    per term, and P filters cost one pass as well (pgh_mat_gemm) -- what a tuner that probes hundreds of coefficient
    vectors on the same personalization (autotune/parameterized.py:94-145) needs."""
    WIDTH = 64
    MAX_TERMS = 4096          # 64 slabs; beyond that the step-by-step route takes over

    def __init__(self, graph, p, chebyshev=False):
        self.graph = graph
        self.n = len(p)
        self.slabs = []
        self.count = 0
        self.l1, self.linf = [], []
        self._last = None
        self.chebyshev = bool(chebyshev)
        self._p = p
        if not self.chebyshev:
            self._push(p)

    def _push(self, col):
        from pygrank_amd.device import DeviceMatrix
        slab, at = divmod(self.count, self.WIDTH)
        if slab == len(self.slabs):
            self.slabs.append(DeviceMatrix.empty(self.n, self.WIDTH))
        self.slabs[slab].set_column(at, col)
        self.l1.append(float(col.abssum()))
        self.linf.append(float(backend.max(backend.abs(col))) if len(col) else 0.0)
        self._last = col
        self.count += 1

    def ensure(self, columns):
        """Makes the first `columns` powers available; False when that is more than a slab set may hold."""
        if columns > self.MAX_TERMS:
            return False
        if self.chebyshev:
            return self._ensure_chebyshev(columns)
        while self.count < columns:
            self._push(self.graph.conv(self._last))
        return True

    def _ensure_chebyshev(self, columns):
        """The terms T_k of the reference's "chebyshev" recurrence (abstract_filters.py:216-224) as slab columns: they come out
        of the engine's f64 recurrence (pgh_poly_terms) 32 at a time -- a term rounded to f32 must not feed the next ones, so an
        extension recomputes from T_1."""
        from pygrank_amd.device import DeviceMatrix
        while self.count < columns:
            slab, at = divmod(self.count, self.WIDTH)
            if slab == len(self.slabs):
                self.slabs.append(DeviceMatrix.empty(self.n, self.WIDTH))
            chunk = min(32, self.WIDTH - at)
            rc = L.lib().pgh_poly_terms(self.graph._h, self._p._h, 1, self.count, chunk, self.slabs[slab]._h, at)
            if rc != 0:
                return False                                 # no blocked f64 image for this graph: the step-by-step route
            for j in range(chunk):
                col = self.slabs[slab].column(at + j)
                self.l1.append(float(col.abssum()))
                self.linf.append(float(backend.max(backend.abs(col))) if len(col) else 0.0)
            self.count += chunk
        return True

    def combine(self, coeffs):
        """sum_k coeffs[k] * power_k as a DeviceVector (coeffs: 1-D, at most `count` long)."""
        c = np.asarray(coeffs, dtype=np.float64)
        out = None
        for slab, first in zip(self.slabs, range(0, len(c), self.WIDTH)):
            part = slab.gemv(c[first:first + self.WIDTH])
            out = part if out is None else out + part
        return out if out is not None else backend.repeat(0.0, self.n)

    def combine_many(self, coeff_matrix):
        """[terms, P] coefficients -> [n, P] slab of results (P <= 64): every slab is read once for all P probes."""
        c = np.asarray(coeff_matrix, dtype=np.float64)
        out = None
        for slab, first in zip(self.slabs, range(0, c.shape[0], self.WIDTH)):
            out = slab.gemm(c[first:first + self.WIDTH], out=out, accumulate=out is not None)
        return out


class ClosedFormGraphFilter(GraphFilter):
    """abstract_filters.py:152-270: sum_k c_k (M^T)^k p with Taylor or the reference's "chebyshev" recurrence."""

    def __init__(self, krylov_dims=None, coefficient_type="taylor", optimization_dict=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if krylov_dims is not None:
            raise NotImplementedError("the Krylov-space approximation (krylov_space.py) is outside the propagation "
                                      "hot path of this build (SURVEY.md 2 row 16)")
        self.krylov_dims = None
        self.coefficient_type = coefficient_type.lower()
        self.optimization_dict = optimization_dict
        self._active_dict = None

    _FORMS = ("taylor", "chebyshev")

    def _start(self, M, personalization, ranks, *args, **kwargs):
        """abstract_filters.py:196-213: the sum starts empty, the running term is the personalization itself."""
        self.coefficient, self.ranks_power = None, personalization.np
        if self.coefficient_type == "chebyshev":
            self.prev_term = 0
        ranks.np = backend.repeat(0.0, backend.length(ranks.np))

    def _recursion(self, result, next_term, next_coefficient):
        """One term of the expansion (abstract_filters.py:215-230) -> (sum so far, term to propagate next).  "taylor": the
        term as it comes.  "chebyshev", as the reference defines it: from the third iteration on the term is 2 * term -
        previous term, the previous term being whatever was propagated at iteration 2 or later; a zero coefficient skips
        the addition (at iteration 2 and before, the reference adds even then)."""
        form = self.coefficient_type
        if form not in self._FORMS:
            raise Exception("Invalid coefficient type")
        it = self.convergence.iteration
        skip_zero = form == "taylor" or it > 2
        if form == "chebyshev" and it >= 2:
            if it > 2:
                next_term = 2 * next_term - self.prev_term
            self.prev_term = next_term
        if skip_zero and self.coefficient == 0:
            return result, next_term
        return result + next_term * next_coefficient, next_term

    def references(self):
        refs = super().references()
        if self.coefficient_type == "chebyshev":
            refs.append("Chebyshev recurrence of the terms")
        if self.optimization_dict is not None:
            refs.append("stored powers of the personalization (optimization dictionary)")
        return refs

    def _prepare(self, personalization):                                          # abstract_filters.py:232-239
        if self.optimization_dict is not None:
            pid = obj2id(personalization)
            if pid not in self.optimization_dict:
                self.optimization_dict[pid] = dict()
            self._active_dict = self.optimization_dict[pid]
        else:
            self._active_dict = None

    def _retrieve_power(self, ranks_power, M):                                    # abstract_filters.py:241-246
        if self._active_dict is not None:
            if self.convergence.iteration not in self._active_dict:
                self._active_dict[self.convergence.iteration] = backend.conv(ranks_power, M)
            return self._active_dict[self.convergence.iteration]
        return backend.conv(ranks_power, M)

    def _step(self, M, personalization, ranks, *args, **kwargs):
        """abstract_filters.py:248-256: draw the next coefficient, add the running term with it, propagate the term once more."""
        weight = self.coefficient = self._coefficient(self.coefficient)
        total, term = self._recursion(ranks.np, self.ranks_power, weight)
        ranks.np = total
        self.ranks_power = self._retrieve_power(term, M)

    def _end(self, M, personalization, ranks, *args, **kwargs):
        """abstract_filters.py:258-267: the per-run attributes of the expansion go away."""
        per_run = ["ranks_power", "coefficient"] + (["prev_term"] if self.coefficient_type == "chebyshev" else [])
        for name in per_run:
            delattr(self, name)
        self._active_dict = None

    def _coefficient(self, previous_coefficient):
        raise Exception("Use a derived class of ClosedFormGraphFilter that implements the _coefficient method")

    def _coefficient_schedule(self, count):
        """c_1 .. c_count exactly as ``_step`` would draw them (``_coefficient`` sees ``convergence.iteration``)."""
        saved = self.convergence.iteration
        coeffs, prev = [], None
        try:
            for it in range(1, count + 1):
                self.convergence.iteration = it
                prev = self._coefficient(prev)
                coeffs.append(float(prev))
        finally:
            self.convergence.iteration = saved
        return coeffs

    def _slab_for(self, M, personalization):
        """The power slab of this personalization inside the active optimisation dict, or None when this configuration
        cannot be served from stored powers."""
        cm = self.convergence
        g = _device_graph(M)
        p = personalization.np
        if type(cm) is not ConvergenceManager or cm.device_error_kind() is None or g is None or g.shape[0] != g.shape[1] \
                or not isinstance(p, DeviceVector) or self.coefficient_type not in self._FORMS or self._active_dict is None:
            return None
        cheb = self.coefficient_type == "chebyshev"
        key = "terms_chebyshev" if cheb else "powers"
        slab = self._active_dict.get(key)
        if not isinstance(slab, _PowerSlab) or slab.graph is not g:
            slab = _PowerSlab(g, p, chebyshev=cheb)
            self._active_dict[key] = slab
        return slab

    def _expansion(self, slab, n):
        """Coefficients of the terms ConvergenceManager (convergence.py:77-101) lets through, decided on the exact change of
        every step, |c_k| * ||term_k|| (the reference compares result_k with result_{k-1} = result_k - c_k term_k).  Returns
        (coeffs, iteration at exit, converged, last change) or None when the slab cannot hold that many powers;
        ``convergence.iteration`` is left as it was."""
        cm = self.convergence
        kind, tol = cm.device_error_kind(), cm.effective_tolerance()
        saved = cm.iteration
        coeffs, prev, it, converged, delta = [], None, 0, False, None
        try:
            while True:
                it += 1
                if it >= cm.max_iters:
                    break
                if delta is not None and kind != L.ERR_ITERS and it % cm.end_modulo == 0 and delta <= tol:
                    converged = True
                    break
                cm.iteration = it                           # _coefficient reads convergence.iteration
                prev = self._coefficient(prev)
                if not slab.ensure(it):
                    return None
                c = float(prev)
                coeffs.append(c)
                norm = slab.linf[it - 1] if kind == L.ERR_LINF else slab.l1[it - 1]
                delta = abs(c) * norm / (n if kind == L.ERR_MABS else 1)
        finally:
            cm.iteration = saved
        return coeffs, it, converged, delta

    def _fused_from_powers(self, M, personalization, ranks, out_scale):
        """optimization_dict route: the filter as ONE pass over the stored powers of this personalization."""
        slab = self._slab_for(M, personalization)
        if slab is None:
            return False
        n = max(len(personalization.np), 1)
        plan = self._expansion(slab, n)
        if plan is None:
            return False                                    # more terms than slabs may hold: the step-by-step route
        coeffs, it, converged, delta = plan
        ranks.np = slab.combine(np.asarray(coeffs, dtype=np.float64) * float(out_scale)) if coeffs \
            else backend.repeat(0.0, len(personalization.np))
        self.last_loop = dict(iterations=it, converged=converged, spmv=0, last_error=delta, loop_ms=0.0, terms=len(coeffs))
        self.convergence.finish_device_loop(it, converged)
        return True

    def rank_many(self, graph, personalization, variants):
        """The ranks of SEVERAL filters of this class on ONE personalization as the columns of an [n, P] device slab
        (P <= 64) -- the probes of an optimiser (autotune/parameterized.py:94-145: every probe differs in the coefficients
        only) or the candidate parameters of a sweep, evaluated in one pass over the stored powers (pgh_mat_gemm).
        `variants`: filters like this one (same graph pipeline, same convergence settings are NOT required: every variant
        stops by its own rule).  Needs an optimisation dict on self (its powers -- or, for the "chebyshev" form, the terms of
        the f64 recurrence -- are shared); returns (DeviceMatrix, [iterations of every variant])."""
        if self.optimization_dict is None:
            raise Exception("rank_many evaluates stored powers: construct the filter with optimization_dict={}")
        if len(variants) < 1 or len(variants) > 64:
            raise Exception("rank_many takes 1 to 64 variants")
        personalization = to_signal(graph, personalization)
        self._prepare(personalization)
        personalization = self.personalization_transform(personalization)
        raw = personalization.np
        norm = raw.abssum() if isinstance(raw, DeviceVector) else backend.sum(backend.abs(raw))
        from pygrank_amd.device import DeviceMatrix
        if norm == 0:
            return DeviceMatrix.from_columns([backend.repeat(0.0, len(raw))] * len(variants)), [0] * len(variants)
        personalization = to_signal(personalization, personalization.np / norm)
        M = self.preprocessor(self._prepare_graph(personalization.graph, personalization))
        slab = self._slab_for(M, personalization)
        if slab is None:
            raise Exception("rank_many needs the engine's graph and a plain ConvergenceManager")
        n = max(len(personalization.np), 1)
        columns, iterations = [], []
        for v in variants:
            if not isinstance(v, ClosedFormGraphFilter) or v.coefficient_type != self.coefficient_type:
                raise Exception("rank_many variants must be closed-form filters of the same coefficient type")
            v.convergence.start()
            plan = v._expansion(slab, n)
            if plan is None:
                raise Exception("a variant needs more powers than the slabs may hold")
            coeffs, it, converged, _ = plan
            v.convergence.finish_device_loop(it, converged)
            scale = norm if v.preserve_norm else 1.0
            columns.append(np.asarray(coeffs, dtype=np.float64) * scale)
            iterations.append(it)
        terms = max((len(c) for c in columns), default=0)
        C_ = np.zeros((max(terms, 1), len(columns)))
        for q, c in enumerate(columns):
            C_[:len(c), q] = c
        return slab.combine_many(C_), iterations

    def _fused_loop(self, M, personalization, ranks, out_scale, *args, **kwargs):
        if args or kwargs or type(self)._step is not ClosedFormGraphFilter._step \
                or type(self)._recursion is not ClosedFormGraphFilter._recursion:
            return False
        if self.optimization_dict is not None:
            return type(self)._retrieve_power is ClosedFormGraphFilter._retrieve_power \
                and self._fused_from_powers(M, personalization, ranks, out_scale)
        if self.coefficient_type not in ("taylor", "chebyshev"):
            raise Exception("Invalid coefficient type")
        cfg = self._loop_cfg(0.0, False, out_scale)
        g = _device_graph(M)
        p, x = personalization.np, ranks.np
        if cfg is None or g is None or g.shape[0] != g.shape[1] or not isinstance(p, DeviceVector) \
                or not isinstance(x, DeviceVector):
            return False
        coeffs = np.asarray(self._coefficient_schedule(max(int(self.convergence.max_iters) - 1, 0)), dtype=np.float64)
        if isinstance(x, LazyVector):
            x = x._writable()              # written in place by the engine
        x._before_write()
        res = L.LoopResult()
        form = 1 if self.coefficient_type == "chebyshev" else 0
        if self._f64_wanted() and _f64_image_usable(g):
            form, cfg = (form or 2), self._f64_cfg(cfg)      # 2: the taylor form with f64 terms and accumulator
        L.check(L.lib().pgh_poly_run(g._h, p._h, coeffs.ctypes.data_as(C.c_void_p), len(coeffs), form, x._h, C.byref(cfg),
                                     C.byref(res)))
        ranks.np = x
        self.last_loop = dict(iterations=res.iterations, converged=bool(res.converged), spmv=res.spmv_count,
                              last_error=res.last_error, loop_ms=res.loop_ms)
        self.convergence.finish_device_loop(res.iterations, res.converged)
        return True


class GenericGraphFilter(ClosedFormGraphFilter):
    """low_pass.py:5-26: filter defined by its hop weights."""

    def __init__(self, weights=None, **kwargs):
        super().__init__(**kwargs)
        self.weights = weights if weights is not None else [0.9] * 10

    def _coefficient(self, _):
        if self.convergence.iteration > len(self.weights):
            return 0
        return self.weights[self.convergence.iteration - 1]

    def _reference(self):
        return f"graph filter with {len(self.weights)} hop weights"


class HeatKernel(ClosedFormGraphFilter):
    """adhoc.py:93-122."""

    def __init__(self, t=3, *args, **kwargs):
        self.t = t
        super().__init__(*args, **kwargs)

    def _coefficient(self, previous_coefficient):                                 # adhoc.py:113-116
        return 1. if previous_coefficient is None else (previous_coefficient * self.t / (self.convergence.iteration + 1))

    def _reference(self):
        return f"heat kernel (t = {self.t:g})"


class PageRankClosed(ClosedFormGraphFilter):
    """adhoc.py:63-90."""

    def __init__(self, alpha=0.85, *args, **kwargs):
        self.alpha = alpha
        super().__init__(*args, **kwargs)

    def _coefficient(self, previous_coefficient):                                 # adhoc.py:83-84
        return 1. if previous_coefficient is None else (previous_coefficient * self.alpha)

    def _reference(self):
        return f"PageRank as a power series (alpha = {self.alpha:g})"


