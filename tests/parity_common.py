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


def runs_in_f64(algo, kwargs):
    """Round 6: a tolerance below fp32 eps sends the engine's whole-loop routes to its f64 image (pygrank_amd/filters.py _f64_wanted), so
    such a case reproduces the REFERENCE's iteration count and result (the golden vectors as they are), not the count of a loop clamped
    at fp32 eps.  The per-step routes (a per-iteration parameter list, a personalization that follows the iterate) stay f32."""
    return not tol_is_fp32_safe(kwargs) and algo != "lowpass" and not kwargs.get("converge_to_eigenvectors")


def partition_personalization(n):
    """The personalization tests/dist_worker.py feeds every rank (ORIGINAL ids)."""
    rng = np.random.default_rng(1)
    p_old = np.zeros(n)
    p_old[rng.choice(n, 20, replace=False)] = rng.random(20) + 0.5
    return p_old


def assemble(parts, perm, key, n):
    """The ranks' slices (new id order) of result `key`, un-permuted into ORIGINAL ids."""
    got = np.zeros(n)
    for part in parts:
        lo, m = int(part["lo"]), int(part["n_local"])
        got[perm[lo:lo + m]] = part[key]
    return got


def check_partition_against_oracle(parts, scale, ef):
    """What tests/dist_worker.py wrote, rank by rank, against the oracle on the un-partitioned graph: PageRank under the three
    stopping rules / without the quotient / from a personalization with negative entries, AbsorbingWalks, HeatKernel,
    PageRankClosed -- <= 1e-6 and equal iteration counts -- and the interleaved run of a Python-driven filter."""
    import scipy.sparse as sp
    from oracle import rmat_np
    n = 1 << scale
    perm = parts[0]["perm"]
    assert sorted(perm.tolist()) == list(range(n))                     # a permutation, identical on every rank
    for part in parts[1:]:
        assert np.array_equal(part["perm"], perm)
    A = rmat_np.rmat_csr(scale, ef, seed=0)
    assert sum(int(part["nnz"]) for part in parts) == A.nnz
    M = sp.csr_array(orc.normalize(A, "col", True))
    p_old = partition_personalization(n)
    runs = {"l1": (p_old, dict(error_type="l1", tol=1e-6, max_iters=500)), "mabs": (p_old, dict(error_type="mabs", tol=1e-7, max_iters=500)),
            "iters": (p_old, dict(error_type="iters", max_iters=21)),
            "noquot": (p_old, dict(error_type="linf", tol=1e-7, max_iters=500, use_quotient=False)),
            "signed": (p_old * np.where(np.arange(n) % 3 == 0, -0.25, 1.0), dict(error_type="l1", tol=1e-6, max_iters=500))}
    for name, (p, kw) in runs.items():
        want, want_iters = orc.pagerank(M, p, alpha=0.85, eps=EPS32, **kw)
        assert all(int(part[name + "_iters"]) == want_iters for part in parts), (name, [int(part[name + "_iters"]) for part in parts], want_iters)
        assert rel_linf(assemble(parts, perm, name + "_ranks", n), want) <= 1e-6, name
    # AbsorbingWalks (adhoc.py:157-169) and the closed-form filters (abstract_filters.py:215-230) on the same partition
    others = (("absorb", lambda: orc.absorbing_walks(M, p_old, alpha=0.85, error_type="l1", tol=1e-6, max_iters=500, eps=EPS32)),
              ("heat", lambda: orc.heat_kernel(M, p_old, t=3, error_type="l1", tol=1e-7, max_iters=100, eps=EPS32)),
              ("heat_mabs", lambda: orc.heat_kernel(M, p_old, t=5, error_type="mabs", tol=1e-9, max_iters=100, eps=EPS32)),
              ("closed", lambda: orc.pagerank_closed(M, p_old, alpha=0.85, error_type="linf", tol=1e-5, max_iters=300, eps=EPS32)))
    for name, ref in others:
        want, want_iters = ref()
        assert all(int(part[name + "_iters"]) == want_iters for part in parts), (name, [int(part[name + "_iters"]) for part in parts], want_iters)
        assert rel_linf(assemble(parts, perm, name + "_ranks", n), want) <= 1e-6, name
    # a Python-driven filter that keeps its buffers, an engine-driven run in between: the same bits before and after
    assert all(int(part["interleaved_equal"]) == 1 for part in parts)
    want_heat = others[1][1]()[0]
    assert all(float(part["interleaved_vs_engine"]) <= 1e-6 * np.max(np.abs(want_heat)) for part in parts)
    return A, M
