"""Shared drivers: run a golden case (tests/golden/cases.py) through pygrank_amd or through the oracle."""
import numpy as np

import cases
from oracle import ref_loops as orc

EPS32 = float(np.finfo(np.float32).eps)
REL_TOL = 1e-6          # BASELINE.json north_star: <= 1e-6 relative L-inf vs the numpy/scipy backend
# The reference's "chebyshev" recurrence S_k = 2 M^T S_{k-1} - S_{k-1} (abstract_filters.py:216-224) amplifies rounding
# noise (x20 on rmat12/heat_cheb): fp32 evaluations of it -- round 1's engine, the reference's own pytorch backend, a numpy
# fp32 restatement -- land between 4e-7 and 1.4e-6.  The engine runs it in f64, so those cases are held to 1e-6 like every
# other.


def tolerance_for(kwargs):
    return REL_TOL


def build_ranker(pg, algo, kwargs):
    kwargs = dict(kwargs)
    kwargs.pop("_absorption", None)
    if "error_type" in kwargs:
        kwargs["error_type"] = {"mabs": pg.Mabs, "l1": pg.L1, "linf": pg.MaxDifference, "iters": "iters"}[kwargs["error_type"]]
    cls = {"pagerank": pg.PageRank, "heat": pg.HeatKernel, "generic": pg.GenericGraphFilter,
           "pagerank_closed": pg.PageRankClosed, "absorbing": pg.AbsorbingWalks,
           "lowpass": pg.LowPassRecursiveGraphFilter, "sarw": pg.SymmetricAbsorbingRandomWalks}[algo]
    return cls(**kwargs)


def run_engine(pg, A, directed, p, algo, kwargs, **ranker_overrides):
    ranker = build_ranker(pg, algo, kwargs)
    for k, v in ranker_overrides.items():
        setattr(ranker, k, v)
    call_kwargs = {}
    if kwargs.get("_absorption") is not None:
        call_kwargs["absorption"] = cases.absorption_vector(kwargs["_absorption"], A.shape[0])
    ranks = ranker.rank(pg.AdjacencyWrapper(A, directed=directed), p.copy(), **call_kwargs)
    return np.asarray(ranks.np, dtype=np.float64), int(ranker.convergence.iteration), ranker


def run_oracle(A, directed, p, algo, kwargs, eps=orc.EPS64):
    kwargs = dict(kwargs)
    absorption = kwargs.pop("_absorption", None)
    pre = {k: kwargs.pop(k) for k in ("normalization", "renormalize") if k in kwargs}
    M = orc.normalize(A, pre.get("normalization", "auto"), directed, pre.get("renormalize", 0.0))
    kwargs["eps"] = eps
    if algo == "pagerank":
        return orc.pagerank(M, p, **kwargs)
    if algo == "heat":
        return orc.heat_kernel(M, p, t=kwargs.pop("t", 3), **kwargs)
    if algo == "generic":
        return orc.generic_filter(M, p, kwargs.pop("weights"), **kwargs)
    if algo == "pagerank_closed":
        return orc.pagerank_closed(M, p, kwargs.pop("alpha"), **kwargs)
    if algo == "absorbing":
        if absorption is not None:
            kwargs["absorption"] = cases.absorption_vector(absorption, A.shape[0])
        return orc.absorbing_walks(M, p, **kwargs)
    if algo == "lowpass":
        return orc.low_pass_recursive(M, p, kwargs.pop("params"), **kwargs)
    if algo == "sarw":
        return orc.symmetric_absorbing_walks(M, p, **kwargs)
    raise KeyError(algo)


def rel_linf(got, want):
    return float(np.max(np.abs(got - want)) / np.max(np.abs(want)))


def tol_is_fp32_safe(kwargs):
    """True when the case's tolerance is not clamped by the engine's fp32 epsilon (convergence.py:101)."""
    if kwargs.get("error_type") == "iters":
        return True
    tol = kwargs.get("tol", 1e-6)
    return tol is not None and tol >= EPS32
