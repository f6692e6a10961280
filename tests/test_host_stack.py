"""CPU tests of the host-side Python (backend module, signals, preprocessor, convergence, filters) driven
through the C-ABI against the host test double (oracle/host_abi.cpp).  The same tests run on the real
engine in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import cases
from parity_common import EPS32, rel_linf, run_engine, run_oracle, runs_in_f64, tol_is_fp32_safe, tolerance_for


@pytest.mark.parametrize("name,gkey,algo,kwargs", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_fused_route_matches_reference(host_engine, golden, graphs, name, gkey, algo, kwargs):
    A, directed, p = graphs(gkey)
    got, iters, ranker = run_engine(host_engine, A, directed, p, algo, kwargs)
    # expectation at the engine's effective tolerance: max(tol, eps_fp32) on the f32 loops (convergence.py:101); the tolerance itself, like
    # the reference's fp64 engine, where a tolerance below fp32 eps sends the run to the f64 image
    f64 = runs_in_f64(algo, kwargs)
    want, want_iters = run_oracle(A, directed, p, algo, kwargs, **({} if f64 else dict(eps=EPS32)))
    assert iters == want_iters
    assert rel_linf(got, want) <= tolerance_for(kwargs)
    if tol_is_fp32_safe(kwargs) or f64:   # then the committed golden vector of the reference applies as-is
        assert iters == int(golden[name + "|iters"])
        assert rel_linf(got, golden[name + "|ranks"]) <= tolerance_for(kwargs)
    if algo != "lowpass" and not kwargs.get("converge_to_eigenvectors"):
        assert hasattr(ranker, "last_loop"), "fused device loop was expected to run"


_GENERIC = [c for c in cases.CASES if c[1] in ("rmat10_dir", "weighted300")]


@pytest.mark.parametrize("name,gkey,algo,kwargs", _GENERIC, ids=[c[0] for c in _GENERIC])
def test_generic_route_matches_reference(host_engine, graphs, name, gkey, algo, kwargs):
    """The fused device loops switched off: the reference's _start / _step / _end structure with one engine call per backend
    primitive -- the route the unmodified reference filters take through the backend module -- held to
    the same bar as the fused route: equal iteration counts, <= 1e-6."""
    A, directed, p = graphs(gkey)
    got, iters, ranker = run_engine(host_engine, A, directed, p, algo, kwargs, _fused_loop=lambda *a, **k: False,
                                    _fused_rank=lambda *a, **k: None)
    assert not hasattr(ranker, "last_loop")
    want, want_iters = run_oracle(A, directed, p, algo, kwargs, eps=EPS32)
    assert iters == want_iters
    assert rel_linf(got, want) <= tolerance_for(kwargs)


def test_converge_to_eigenvectors(host_engine, graphs):
    """RecursiveGraphFilter(converge_to_eigenvectors=True) (abstract_filters.py:135-136: the personalization follows the
    iterate, so the loop converges to the dominant eigenvector whatever the seeds): the reference pins it by rank
    correlation with alpha = 0.99 (tests/test_filter_optimization.py:6-15, Spearman > 0.99); here additionally against
    the oracle's restatement of the same loop."""
    import scipy.stats
    from oracle import ref_loops as orc
    pg = host_engine
    A, directed, p = graphs("rmat12_sym")
    graph = pg.AdjacencyWrapper(A, directed=directed)
    eig = pg.PageRank(0.85, converge_to_eigenvectors=True, tol=1e-9, max_iters=2000)
    r1 = np.asarray(eig.rank(graph, p.copy()).np, dtype=np.float64)
    assert not hasattr(eig, "last_loop")                   # per-step route: the personalization changes every step
    M = orc.normalize(A, "auto", directed)
    want, want_iters = orc.pagerank(M, p, alpha=0.85, converge_to_eigenvectors=True, tol=1e-9, max_iters=2000, eps=EPS32)
    assert eig.convergence.iteration == want_iters
    assert rel_linf(r1, want) <= 1e-6
    r2 = np.asarray(pg.PageRank(0.99, tol=1e-9, max_iters=5000).rank(graph, p.copy()).np, dtype=np.float64)
    assert scipy.stats.spearmanr(r1, r2)[0] > 0.99
