"""Engine-agnostic check bodies (take the loaded ``pygrank_amd`` module).  Each restates assertions of the
reference's own tests; file:line given per check."""
import networkx as nx
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ref_loops as orc


def _graph9():
    """Stand-in for the reference's downloaded 'graph9': 9 nodes A..I, undirected."""
    G = nx.Graph()
    G.add_edges_from([("A", "B"), ("A", "C"), ("B", "C"), ("C", "D"), ("D", "E"), ("E", "F"), ("F", "G"),
                      ("G", "H"), ("H", "I"), ("D", "F"), ("B", "E")])
    return G


def check_primitive_conversion(pg):                          # tests/test_core.py:25-33
    assert pg.obj2id("str") == str(hash("str"))
    assert pg.sum(pg.to_array([1, 2, 3])) == 6
    assert pg.sum(pg.dot(pg.exp(pg.log(pg.to_array([4, 5]))), pg.to_array([2, 2]))) == pytest.approx(18, rel=1e-6)
    primitive = pg.to_array([1, 2, 3])
    assert id(primitive) == id(pg.to_array(primitive, copy_array=False))
    assert id(primitive) != id(pg.to_array(primitive, copy_array=True))


def check_separate_and_combine(pg):                          # tests/test_core.py:36-44
    table = pg.to_primitive([[1, 2, 3], [4, 5, 6]])
    cols = pg.separate_cols(table)
    assert len(cols) == 3
    for col in cols:
        assert pg.length(col) == 2
    new_table = pg.combine_cols(cols)
    assert pg.sum(pg.abs(table - new_table)) == 0


def check_signal_init(pg):                                   # tests/test_core.py:74-83
    with pytest.raises(Exception):
        pg.GraphSignal([1, 2, 3], [1, 2])
    signal = pg.GraphSignal(_graph9(), {"A": 1, "B": 2})
    del signal["A"]
    assert signal["A"] == 0
    assert signal["B"] == 2


def check_backend_load(pg):                                  # tests/test_core.py:91-101
    assert pg.backend_name() == "hip"
    with pytest.raises(Exception):
        pg.load_backend("unknown")
    with pytest.raises(Exception):
        pg.load_backend("numpy")          # this build ships exactly one engine
    assert pg.backend_name() == "hip"
    with pg.Backend("hip") as backend:
        assert backend.backend_name() == "hip"
    assert pg.backend_name() == "hip"


def check_signal_direct_operations(pg):                      # tests/test_core.py:122-155
    graph = nx.DiGraph([(1, 2), (2, 3)])
    signal = pg.to_signal(graph, [1., 2., 3.])
    assert pg.sum(signal) == 6
    assert pg.sum(signal + 1) == 9
    assert pg.sum(1 + signal) == 9
    assert pg.sum(signal ** 2) == 14
    assert pg.sum(signal - pg.to_signal(graph, [1, 2, 2])) == 1
    assert pg.sum(-1 + signal) == 3
    assert pg.sum(signal / pg.to_signal(graph, [1., 2., 3.])) == 3
    assert pg.sum(3 ** signal) == 3 + 9 + 27
    signal **= 2
    assert pg.sum(signal) == 14
    signal.np = pg.to_signal(graph, [4, 4, 4])
    assert pg.sum(signal) == 12
    assert pg.sum(+signal) == 12
    assert pg.sum(-signal) == -12
    assert pg.sum(-signal / 2) == -6
    assert pg.sum(2 / signal) == 1.5
    signal += 1
    assert pg.sum(signal) == 15
    signal -= 1
    assert pg.sum(signal) == 12
    signal /= 2
    assert pg.sum(signal) == 6
    signal /= 2
    assert pg.sum(signal) == 3
    signal *= 4
    assert pg.sum(signal) == 12
    with pytest.raises(Exception):
        signal + pg.to_signal(graph.copy(), [1., 2., 3.])


def check_vector_protocol(pg):                               # SURVEY.md 8a row a4 operator protocol
    x = pg.to_array([3., -1., 0., 2.])
    y = pg.to_array([1., 1., 2., 2.])
    assert np.allclose(np.asarray(x > y), [1, 0, 0, 0])
    assert np.allclose(np.asarray(x != y), [1, 1, 1, 0])
    assert np.allclose(np.asarray(x == y), [0, 0, 0, 1])
    assert np.allclose(np.asarray(pg.filter_out(x, pg.to_array([0., 1., 0., 1.]))), [3., 0.])
    assert np.allclose(np.asarray(x[x > 0]), [3., 2.])
    assert float(x[1]) == -1.0
    x[1] = 5.0
    assert float(x[1]) == 5.0
    assert pg.max(x) == 5 and pg.min(x) == 0 and pg.mean(x) == 2.5
    assert pg.length(x) == 4 and pg.is_array(x) and pg.is_array([1]) and not pg.is_array(3.0)
    assert np.allclose(np.asarray(pg.repeat(0.5, 3)), [0.5] * 3)
    z = pg.self_normalize(pg.to_array([1., -1., 2.]))
    assert np.allclose(np.asarray(z), [0.25, -0.25, 0.5])
    assert np.allclose(np.asarray(pg.safe_inv(pg.to_array([2., 0., 4.]))), [0.5, 0., 0.25])
    assert pg.epsilon() == float(np.finfo(np.float32).eps)
    assert pg.sum(pg.abs(pg.to_array([]))) == 0
    assert np.allclose(np.asarray(pg.to_array(np.ones((3, 1)))), [1, 1, 1])     # (n, 1) is flattened
    assert pg.cast(x) is x and pg.graph_dropout("M", 0) == "M"


def check_preprocessor_types(pg):                            # tests/test_preprocessor.py:6-15
    graph = _graph9()
    rng = np.random.default_rng(0)
    signal = pg.to_signal(graph, {v: rng.random() for v in graph})
    laplacian = pg.preprocessor(normalization="laplacian")(graph)
    symmetric = pg.preprocessor(normalization="symmetric")(graph)
    assert abs(pg.sum(pg.conv(signal, laplacian) + pg.conv(signal, symmetric) - signal)) <= 4 * pg.epsilon()


def check_preprocessor_hashing(pg):                          # tests/test_preprocessor.py:18-46
    with pytest.raises(Exception):
        pg.preprocessor(normalization="unknown", assume_immutability=True)(_graph9())
    pre = pg.preprocessor(normalization="col", assume_immutability=False)
    graph = _graph9()
    assert id(pre(graph)) != id(pre(graph))
    pre = pg.MethodHasher(pg.preprocessor, assume_immutability=True)
    graph = _graph9()
    res1 = pre(graph)
    pre.assume_immutability = False
    assert id(res1) != id(pre(graph))
    pre = pg.preprocessor(normalization="col", assume_immutability=True)
    graph = _graph9()
    res1 = pre(graph)
    assert id(res1) == id(pre(graph))
    pre.clear_hashed()
    assert id(res1) != id(pre(graph))


def check_upload_matches_reference_normalisation(pg):        # preprocessing.py:99-144 + numpy.py:76-77
    rng = np.random.default_rng(3)
    A = sp.random(60, 60, density=0.1, random_state=np.random.RandomState(5), format="csr")
    A = sp.csr_array(A)
    for normalization in ["col", "symmetric", "both", "laplacian", "none"]:
        for renorm in [False, True]:
            M = orc.normalize(A, normalization, True, float(renorm))
            adj = pg.preprocessor(normalization=normalization, renormalize=renorm)(pg.AdjacencyWrapper(A, directed=True))
            got = adj.array.download_transposed()
            assert np.allclose(got.toarray(), M.T.toarray(), rtol=2e-7, atol=1e-12)
            assert np.allclose(np.asarray(pg.degrees(adj)), orc.row_sums(M), rtol=2e-7, atol=1e-7)
            x = rng.random(60)
            assert np.allclose(np.asarray(pg.conv(pg.to_array(x), adj)), x @ M, rtol=2e-6, atol=1e-7)


def check_zero_personalization(pg):                          # tests/test_filters.py:9-10
    assert pg.sum(pg.PageRank()(_graph9(), {}).np) == 0


def check_abstract_filter_types(pg):                         # tests/test_filters.py:13-20
    graph = _graph9()
    for cls in (pg.GraphFilter, pg.RecursiveGraphFilter, pg.ClosedFormGraphFilter):
        with pytest.raises(Exception):
            cls().rank(graph)


def check_invalid_parameters(pg):                            # tests/test_filters.py:24-29, test_core.py:21-22
    graph = _graph9()
    with pytest.raises(Exception):
        pg.HeatKernel(normalization="unknown").rank(graph)
    with pytest.raises(Exception):
        pg.HeatKernel(coefficient_type="unknown").rank(graph)
    with pytest.raises(Exception):
        pg.PageRank(krylov_dims=5)
    with pytest.raises(Exception):
        pg.PageRank().rank(list(graph))                      # tests/test_filters.py:53-56


def check_convergence_string(pg):                            # tests/test_filters.py:32-38
    ranker = pg.PageRank() >> pg.Normalize()
    ranker(_graph9())
    assert str(ranker.convergence.iteration) + " iterations" in str(ranker.convergence)


def check_pagerank_vs_networkx(pg):                          # tests/test_filters.py:41-50
    graph = _graph9().to_directed()
    ranker = pg.Normalize("sum", pg.PageRank(normalization="col", tol=1e-9, max_iters=1000))
    want = nx.pagerank(graph, tol=1e-12)
    got = ranker(graph)
    assert max(abs(got[v] - want[v]) for v in graph) < 5e-7   # fp32 engine: eps-level agreement in fp32


def check_non_convergence(pg):                               # tests/test_filters.py:59-62
    with pytest.raises(Exception):
        pg.PageRank(max_iters=5).rank(_graph9())


def check_custom_runs(pg):                                   # tests/test_filters.py:65-72
    graph = _graph9()
    tol = pg.epsilon()
    ranks1 = pg.Normalize(pg.PageRank(0.85, tol=tol, max_iters=1000, use_quotient=False)).rank(graph, {"A": 1})
    ranks2 = pg.Normalize(pg.GenericGraphFilter([0.85 ** i * len(graph) for i in range(80)], tol=tol)).rank(graph, {"A": 1})
    ranks3 = pg.Normalize(pg.LowPassRecursiveGraphFilter([0.85 for _ in range(80)], tol=tol)).rank(graph, {"A": 1})
    assert pg.Mabs(ranks1)(ranks2) < 1.E-6
    assert pg.Mabs(ranks1)(ranks3) < 1.E-6


def check_stream(pg):                                        # tests/test_filters.py:85-98
    graph = _graph9()
    ranks1 = pg.Normalize(pg.PageRank(0.85, tol=pg.epsilon(), max_iters=1000, use_quotient=False)).rank(graph, {"A": 1})
    ranks2 = pg.to_signal(graph, {"A": 1}) >> pg.PageRank(0.85, tol=pg.epsilon(), max_iters=1000) + pg.Tautology() >> pg.Normalize()
    assert pg.Mabs(ranks1)(ranks2) < 4 * pg.epsilon()
    ranks1 = pg.GenericGraphFilter([0, 0, 1], max_iters=4, error_type="iters") | pg.to_signal(graph, {"A": 1})
    ranks2 = pg.GenericGraphFilter([1, 1, 1], tol=None) & ~pg.GenericGraphFilter([1, 1], tol=None) | pg.to_signal(graph, {"A": 1})
    assert pg.Mabs(ranks1)(ranks2) < 4 * pg.epsilon()


def check_quotient(pg):                                      # tests/test_filters.py:118-135
    graph = _graph9()
    tol = max(1.E-9, pg.epsilon())
    a = pg.PageRank(normalization="symmetric", tol=tol, use_quotient=True).rank(graph)
    b = pg.PageRank(normalization="symmetric", tol=tol, use_quotient=pg.Normalize("sum")).rank(graph)
    assert pg.Mabs(a)(b) < 4 * pg.epsilon()
    c = pg.Normalize(pg.PageRank(normalization="symmetric", tol=tol, use_quotient=True)).rank(graph)
    d = pg.PageRank(tol=tol) + pg.preprocessor(normalization="symmetric") + pg.Normalize("sum") >> pg.Normalize() \
        | pg.to_signal(graph, {v: 1 for v in graph})
    assert pg.Mabs(c)(d) < 4 * pg.epsilon()


def check_automatic_graph_casting(pg):                       # tests/test_filters.py:138-148
    graph = _graph9()
    signal = pg.to_signal(graph, {"A": 1})
    r1 = pg.PageRank(normalization="col").rank(signal, signal)
    r2 = pg.PageRank(normalization="col").rank(personalization=signal)
    assert pg.Mabs(r1)(r2) < pg.epsilon()
    with pytest.raises(Exception):
        pg.PageRank(normalization="col").rank(personalization={"A": 1})
    with pytest.raises(Exception):
        pg.PageRank(normalization="col").rank(graph.copy(), signal)


def check_absorbing_vs_pagerank(pg):                         # tests/test_filters.py:151-157
    graph = _graph9()
    p = {"A": 1, "B": 1}
    a = pg.PageRank(normalization="col").rank(graph, p)
    b = pg.AbsorbingWalks(0.85, normalization="col", max_iters=1000).rank(graph, p)
    assert pg.Mabs(a)(b) < 4 * pg.epsilon()


def check_lowpass_vs_pagerank(pg):                           # tests/test_filters.py:160-166
    graph = _graph9()
    p = {"A": 1, "B": 1}
    a = pg.PageRank(0.9, use_quotient=False, max_iters=11, error_type="iters").rank(graph, p)
    b = pg.LowPassRecursiveGraphFilter().rank(graph, p)
    assert pg.Mabs(a)(b) < 4 * pg.epsilon()


def check_kernel_locality(pg):                               # tests/test_filters.py:169-177
    graph = _graph9()
    p = {"A": 1, "B": 1}
    a = pg.Normalize("sum", pg.PageRank(max_iters=1000)).rank(graph, p)
    b = pg.Normalize("sum", pg.HeatKernel(max_iters=1000)).rank(graph, p)
    assert a["A"] < b["A"]
    assert a["I"] > b["I"]


def check_optimization_dict(pg):                             # tests/test_filters.py:180-197 (cache semantics)
    graph = _graph9()
    pre = pg.preprocessor(assume_immutability=True)
    signal = pg.to_signal(graph, {"A": 1, "B": 1})
    cache = dict()
    a = pg.HeatKernel(t=3, preprocessor=pre, optimization_dict=cache, error_type="iters", max_iters=12).rank(signal)
    inner = next(iter(cache.values()))
    # the reference keeps 11 cached convolutions (one per iteration); the engine keeps the 11 powers it needs as the
    # columns of one device slab (filters._PowerSlab)
    assert len(cache) == 1 and (len(inner) == 11 or inner["powers"].count == 11)
    b = pg.HeatKernel(t=3, preprocessor=pre, optimization_dict=cache, error_type="iters", max_iters=12).rank(signal)
    c = pg.HeatKernel(t=3, preprocessor=pre, error_type="iters", max_iters=12).rank(signal)
    assert pg.Mabs(a)(b) == 0
    assert pg.Mabs(a)(c) < 4 * pg.epsilon()


def check_power_slab_serves_tuner_probes(pg):
    """with an optimisation dict the powers {(M^T)^k p} of a personalization live in device slabs and every
    further filter on that personalization (a tuner probing weight vectors, autotune/parameterized.py:135-145) is one pass
    over them.  Checked against the ORACLE's restatement of the reference loop (oracle/ref_loops.py: generic_filter /
    heat_kernel / pagerank_closed -- abstract_filters.py:196-256), not against the engine's own step-by-step route: ranks to
    1e-6, equal iteration counts, no further convolutions."""
    import cases
    from oracle import ref_loops as orc
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    M = orc.normalize(A, "auto", directed)
    eps32 = float(np.finfo(np.float32).eps)
    pre = pg.preprocessor(assume_immutability=True)
    signal = pg.to_signal(graph, p.copy())
    cache = dict()
    rng = np.random.default_rng(4)
    probes = [list(rng.random(12)) for _ in range(5)] + [[0.5, 0.3, 0, 0.2], [1.0]]
    columns = None
    for weights in probes:
        cached = pg.GenericGraphFilter(weights, preprocessor=pre, optimization_dict=cache, tol=1e-8, max_iters=100)
        got = np.asarray(cached.rank(signal).np, dtype=np.float64)
        want, want_iters = orc.generic_filter(M, p, weights, tol=1e-8, max_iters=100, eps=eps32)
        assert cached.convergence.iteration == want_iters, (weights, cached.convergence.iteration, want_iters)
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), weights
        assert cached.last_loop["spmv"] == 0 and cached.last_loop["terms"] == cached.convergence.iteration - 1
        slab = next(iter(cache.values()))["powers"]
        columns = max(columns or 0, slab.count)
        assert slab.count == columns                      # the slabs only ever grow to the longest probe
    assert len(cache) == 1 and columns <= 14
    # HeatKernel to a tolerance through the same slab machinery, and the non-convergence exception
    hk = pg.HeatKernel(3, preprocessor=pre, optimization_dict=dict(), tol=1e-7, max_iters=60)
    a = np.asarray(hk.rank(signal).np, dtype=np.float64)
    b, b_iters = orc.heat_kernel(M, p, t=3, tol=1e-7, max_iters=60, eps=eps32)
    assert hk.convergence.iteration == b_iters and np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(b))
    try:
        pg.HeatKernel(3, preprocessor=pre, optimization_dict=dict(), tol=1e-12, max_iters=5).rank(signal)
        raise AssertionError("expected a non-convergence exception")
    except Exception as exc:
        assert "converge" in str(exc)
    # expansions of MORE than one slab's 64 powers (the route used to leave convergence.iteration behind when it
    # gave up at 65 terms): PageRankClosed alpha = 0.9 needs ~150 terms for 1e-7, HeatKernel t = 30 ~90
    for make, ref in ((lambda **kw: pg.PageRankClosed(0.9, preprocessor=pre, tol=1e-7, max_iters=400, **kw),
                       lambda: orc.pagerank_closed(M, p, alpha=0.9, tol=1e-7, max_iters=400, eps=eps32)),
                      (lambda **kw: pg.HeatKernel(30, preprocessor=pre, tol=1e-7, max_iters=400, **kw),
                       lambda: orc.heat_kernel(M, p, t=30, tol=1e-7, max_iters=400, eps=eps32))):
        long_cache = dict()
        algo = make(optimization_dict=long_cache)
        got = np.asarray(algo.rank(signal).np, dtype=np.float64)
        want, want_iters = ref()
        assert want_iters > 66 and algo.convergence.iteration == want_iters, (algo.convergence.iteration, want_iters)
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))
        assert algo.last_loop["spmv"] == 0 and len(next(iter(long_cache.values()))["powers"].slabs) >= 2
        again = make(optimization_dict=long_cache)      # second probe: served from the slabs, same answer
        assert np.array_equal(np.asarray(again.rank(signal).np), np.asarray(algo.rank(signal).np))


def check_chebyshev_slab_serves_tuner_probes(pg):
    """SURVEY.md 8f-2 for the reference's "chebyshev" coefficient type (abstract_filters.py:216-224): the terms T_k of the
    recurrence come out of the engine's f64 route once (pgh_poly_terms) and live as slab columns; every further filter of that
    form on the personalization -- single probes and rank_many -- is a pass over them.  Against the oracle's loop: ranks to
    1e-6, equal iteration counts, no further convolutions."""
    import cases
    from oracle import ref_loops as orc
    eps32 = float(np.finfo(np.float32).eps)
    for gkey in ("rmat10_dir", "rmat12_sym"):
        A, directed, p = cases.GRAPHS[gkey]()
        graph = pg.AdjacencyWrapper(A, directed=directed)
        M = orc.normalize(A, "auto", directed)
        pre = pg.preprocessor(assume_immutability=True)
        signal = pg.to_signal(graph, p.copy())
        cache = dict()
        rng = np.random.default_rng(6)
        probes = [list(rng.random(10)) for _ in range(3)] + [[1, 0.5, 0, 0.25, 0.1]]
        for weights in probes:
            algo = pg.GenericGraphFilter(weights, coefficient_type="chebyshev", preprocessor=pre, optimization_dict=cache, tol=1e-8, max_iters=60)
            got = np.asarray(algo.rank(signal).np, dtype=np.float64)
            want, want_iters = orc.generic_filter(M, p, weights, coefficient_type="chebyshev", tol=1e-8, max_iters=60, eps=eps32)
            assert algo.convergence.iteration == want_iters, (gkey, weights, algo.convergence.iteration, want_iters)
            assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), (gkey, weights)
            assert algo.last_loop["spmv"] == 0
        slab = next(iter(cache.values()))["terms_chebyshev"]
        assert slab.chebyshev and slab.count == 32 and len(cache) == 1
        hk = pg.HeatKernel(5, coefficient_type="chebyshev", preprocessor=pre, optimization_dict=cache, error_type="iters", max_iters=31)
        a = np.asarray(hk.rank(signal).np, dtype=np.float64)
        b, b_iters = orc.heat_kernel(M, p, t=5, coefficient_type="chebyshev", error_type="iters", max_iters=31)
        assert hk.convergence.iteration == b_iters == 31 and np.max(np.abs(a - b)) <= 1e-6 * np.max(np.abs(b)), gkey
        # many probes in one pass over the terms
        variants = [pg.GenericGraphFilter(w, coefficient_type="chebyshev", preprocessor=pre, tol=1e-8, max_iters=60) for w in probes]
        slab_out, iterations = variants[0].__class__(probes[0], coefficient_type="chebyshev", preprocessor=pre, optimization_dict=cache, tol=1e-8,
                                                     max_iters=60).rank_many(graph, p.copy(), variants)
        cols = np.asarray(slab_out)
        for q, weights in enumerate(probes):
            want, want_iters = orc.generic_filter(M, p, weights, coefficient_type="chebyshev", tol=1e-8, max_iters=60, eps=eps32)
            assert iterations[q] == want_iters and np.max(np.abs(cols[:, q] - want)) <= 1e-6 * np.max(np.abs(want)), (gkey, q)


def check_rank_many_probes_in_one_pass(pg):
    """SURVEY.md 8f-2 "many probes as one GEMM": P coefficient vectors on one personalization -> an [n, P] slab from ONE pass
    over the stored powers (pgh_mat_gemm); every column against the oracle's loop for that probe (abstract_filters.py:
    196-256 as restated in oracle/ref_loops.py), with that probe's own stopping iteration."""
    import cases
    from oracle import ref_loops as orc
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    M = orc.normalize(A, "auto", directed)
    eps32 = float(np.finfo(np.float32).eps)
    pre = pg.preprocessor(assume_immutability=True)
    rng = np.random.default_rng(9)
    weight_sets = [list(rng.random(int(k))) for k in rng.integers(3, 40, size=21)] + [[1.0], [0.0, 0.0, 1.0]]
    cache = dict()
    head = pg.GenericGraphFilter(weight_sets[0], preprocessor=pre, optimization_dict=cache, tol=1e-8, max_iters=100)
    variants = [pg.GenericGraphFilter(w, preprocessor=pre, tol=1e-8, max_iters=100) for w in weight_sets]
    slab, iterations = head.rank_many(graph, p.copy(), variants)
    got = np.asarray(slab.numpy() if hasattr(slab, "numpy") else slab, dtype=np.float64)
    assert got.shape == (len(p), len(weight_sets))
    for q, w in enumerate(weight_sets):
        want, want_iters = orc.generic_filter(M, p, w, tol=1e-8, max_iters=100, eps=eps32)
        assert iterations[q] == want_iters, (q, iterations[q], want_iters)
        assert np.max(np.abs(got[:, q] - want)) <= 1e-6 * np.max(np.abs(want)), q
    # heat kernels of different t through the same powers (a parameter sweep), one of them beyond 64 terms
    ts = [1.0, 3.0, 5.0, 30.0]
    hk = [pg.HeatKernel(t, preprocessor=pre, tol=1e-7, max_iters=400) for t in ts]
    slab, iterations = head.rank_many(graph, p.copy(), hk)
    got = np.asarray(slab.numpy(), dtype=np.float64)
    for q, t in enumerate(ts):
        want, want_iters = orc.heat_kernel(M, p, t=t, tol=1e-7, max_iters=400, eps=eps32)
        assert iterations[q] == want_iters and np.max(np.abs(got[:, q] - want)) <= 1e-6 * np.max(np.abs(want)), t


def check_device_postprocessors_and_measures(pg):
    """SURVEY.md 8f-3 against outcomes of the reference itself (tests/golden/golden_post.npz, written by make_golden.py):
    Ordinals / Top / Threshold are exact on the reference's own ranks (integer / 0-1 outputs of one device sort and
    elementwise kernels); Sweep / LinearSweep / Transformer / Normalize and the residual-style measures to fp32 rounding."""
    import os
    import cases
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_post.npz"))
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    ref_ranks = pg.to_signal(graph, gold["ranks"])
    f32 = gold["ranks"].astype(np.float32).astype(np.float64)
    distinct = len(np.unique(f32)) == len(np.unique(gold["ranks"]))       # no ties created by the fp32 storage

    def out(algo):
        return np.asarray(algo.transform(ref_ranks).np, dtype=np.float64)
    if distinct:
        assert np.array_equal(out(pg.Ordinals()), gold["post|ordinals"])
    else:                                                                  # ties may swap neighbours: same multiset of ranks
        assert np.array_equal(np.sort(out(pg.Ordinals())), np.sort(gold["post|ordinals"]))
    assert np.array_equal(out(pg.Top(5)), gold["post|top5"])
    assert np.array_equal(out(pg.Top(0.5)), gold["post|top_half"])
    assert np.array_equal(out(pg.Threshold(0.02)), gold["post|threshold"])
    assert np.array_equal(out(pg.Threshold(0.02, inclusive=True)), gold["post|threshold_inclusive"])
    assert np.array_equal(out(pg.Threshold("gap")), gold["post|threshold_gap"])
    for key, algo in (("transformer_exp", pg.Transformer()), ("normalize_range", pg.Normalize("range")), ("normalize_l2", pg.Normalize("L2"))):
        got = out(algo)
        assert np.max(np.abs(got - gold["post|" + key])) <= 4e-7 * np.max(np.abs(gold["post|" + key])), key
    # the sweeps run the whole pipeline (two PageRank runs on the engine)
    base = lambda: pg.PageRank(0.85, error_type="iters", max_iters=41)       # noqa: E731  (a stopping rule fp32 and fp64 engines share)
    for key, algo in (("sweep", pg.Sweep(base())), ("linear_sweep", pg.LinearSweep(base()))):
        got = np.asarray(algo.rank(graph, p.copy()).np, dtype=np.float64)
        want = gold["post|" + key]
        assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want)), (key, np.max(np.abs(got - want)) / np.max(np.abs(want)))
    # AUC with one device sort against sklearn's values (ties at their mid-rank; an f32 copy of the scores may tie what f64 kept apart)
    labels = pg.to_array(gold["auc|labels"])
    for key, scores in (("ranks", gold["ranks"]), ("coarse", gold["auc|coarse_scores"]), ("random", gold["measure|u"])):
        got, want = float(pg.AUC(labels)(pg.to_array(scores))), float(gold["auc|" + key])
        assert abs(got - want) <= 2e-6, (key, got, want)
    import pytest
    with pytest.raises(Exception, match="all labels are the same"):
        pg.AUC(pg.to_array(np.ones(len(gold["ranks"]))))(pg.to_array(gold["ranks"]))
    u, v = gold["measure|u"], gold["measure|v"]
    du, dv = pg.to_array(u), pg.to_array(v)
    for key, m in (("rmabs", pg.RMabs), ("msq", pg.MSQ), ("msqrt", pg.MSQRT), ("l2", pg.L2), ("euclidean", pg.Euclidean),
                   ("cos", pg.Cos), ("dot", pg.Dot)):
        got, want = float(m(du)(dv)), float(gold["measure|" + key])
        assert abs(got - want) <= 1e-6 * abs(want), (key, got, want)


def check_generic_route_equals_fused_route(pg):
    """The per-step backend-primitive route (reference structure) and the fused device loop agree."""
    import cases
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    for make in (lambda **k: pg.PageRank(0.85, tol=1e-6, **k),
                 lambda **k: pg.PageRank(0.85, use_quotient=False, error_type=pg.L1, tol=1e-6, max_iters=500, **k),
                 lambda **k: pg.AbsorbingWalks(0.9, tol=1e-6, max_iters=500, **k),
                 lambda **k: pg.HeatKernel(t=5, error_type="iters", max_iters=21, **k),
                 lambda **k: pg.HeatKernel(t=5, coefficient_type="chebyshev", error_type="iters", max_iters=21, **k),
                 lambda **k: pg.GenericGraphFilter([0.5, 0.3, 0, 0.2], tol=1e-7, **k)):
        fused = make()
        r_fused = fused.rank(graph, p.copy())
        assert hasattr(fused, "last_loop")
        generic = make()
        generic._fused_loop = lambda *a, **k: False
        r_generic = generic.rank(graph, p.copy())
        assert generic.convergence.iteration == fused.convergence.iteration
        got, want = np.asarray(r_fused.np), np.asarray(r_generic.np)
        assert np.max(np.abs(got - want)) <= 2e-6 * np.max(np.abs(want))


def check_warm_start_and_propagate(pg):                      # abstract_filters.py:47,56; signals.py:225-226
    import cases
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    base = pg.PageRank(0.85, tol=1e-7, max_iters=500)
    r = base.rank(graph, p.copy())
    iters_cold = base.convergence.iteration
    warm = pg.PageRank(0.85, tol=1e-7, max_iters=500)
    r2 = warm.rank(graph, p.copy(), warm_start=np.asarray(r.np) / np.asarray(r.np).sum())
    assert warm.convergence.iteration < iters_cold
    assert np.max(np.abs(np.asarray(r2.np) - np.asarray(r.np))) < 1e-5 * np.asarray(r.np).max()
    feats = np.stack([p, np.roll(p, 7)], axis=1)
    out = pg.PageRank(0.85).propagate(graph, pg.to_primitive(feats))
    cols = np.asarray(out)
    assert cols.shape == (A.shape[0], 2)
    assert np.allclose(cols[:, 0], np.asarray(pg.PageRank(0.85).rank(graph, p.copy()).np), rtol=1e-6, atol=1e-9)


def check_pagerank_float64_storage(pg):
    """The reference's numpy backend is fp64: epsilon() = finfo(float64).eps (pygrank/core/backend/numpy.py:84-86), and its tests run
    tol = 1e-9 (tests/test_filters.py:189,194).  The f32 loop clamps that tolerance at fp32 eps; PageRank(dtype="float64") keeps the loop
    in f64 (pgh_ppr_run_f64) and stops where the REFERENCE stops: the committed golden runs (tests/golden/golden.npz, made by the
    reference itself) -- er10k: 18 iterations -- and the oracle at fp64 eps, ranks <= 1e-6 (they leave as f32)."""
    import os
    import cases
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz"))
    for name, gkey, algo, kwargs in cases.CASES:
        if algo != "pagerank" or kwargs.get("tol", 1) > 2e-9 or kwargs.get("converge_to_eigenvectors"):
            continue
        A, directed, p = cases.GRAPHS[gkey]()
        kw = {k: v for k, v in kwargs.items() if k not in ("normalization", "renormalize")}
        pre = pg.preprocessor(normalization=kwargs.get("normalization", "auto"), renormalize=kwargs.get("renormalize", False))
        ranker = pg.PageRank(preprocessor=pre, dtype="float64", **kw)
        got = np.asarray(ranker.rank(pg.AdjacencyWrapper(A, directed=directed), p.copy()).np, dtype=np.float64)
        want_iters = int(golden[name + "|iters"])
        assert ranker.convergence.iteration == want_iters, (name, ranker.convergence.iteration, want_iters)
        want = golden[name + "|ranks"]
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), name
        # the f32 loop cannot honour that tolerance (it stops at fp32 eps): fewer iterations than the reference
        f32 = pg.PageRank(preprocessor=pre, dtype="float32", **kw)      # (dtype=None picks f64 by itself for such a tolerance: round 6)
        f32.rank(pg.AdjacencyWrapper(A, directed=directed), p.copy())
        assert f32.convergence.iteration < want_iters, name
    # warm start, the max rule without the quotient, a fixed step count: against the oracle at fp64 eps
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    M = orc.normalize(A, "auto", directed)
    graph = pg.AdjacencyWrapper(A, directed=directed)
    for kw_engine, kw_oracle in ((dict(error_type=pg.MaxDifference, tol=1e-11, max_iters=500, use_quotient=False), dict(error_type="linf", tol=1e-11, max_iters=500, use_quotient=False)),
                                 (dict(error_type="iters", max_iters=30), dict(error_type="iters", max_iters=30)),
                                 (dict(error_type=pg.L1, tol=1e-10, max_iters=500), dict(error_type="l1", tol=1e-10, max_iters=500))):
        ranker = pg.PageRank(0.85, dtype="float64", **kw_engine)
        got = np.asarray(ranker.rank(graph, p.copy()).np, dtype=np.float64)
        want, want_iters = orc.pagerank(M, p, alpha=0.85, **kw_oracle)
        assert ranker.convergence.iteration == want_iters, (kw_oracle, ranker.convergence.iteration, want_iters)
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), kw_oracle
    with pytest.raises(Exception):
        pg.PageRank(0.85, dtype="float16")


def check_f64_iterates_for_every_filter(pg):
    """VERDICT r5 item 6.  The reference's numpy engine is fp64 throughout (pygrank/core/backend/numpy.py:84-86); an f32 loop clamps a
    tolerance below fp32 eps (convergence.py:101) and stops early.  With dtype=None a tolerance below fp32 eps sends PageRank,
    AbsorbingWalks, SymmetricAbsorbingRandomWalks and the closed-form filters (taylor and "chebyshev") to f64 iterates on the engine's
    f64 image: the REFERENCE's iteration counts (its own golden runs: AbsorbingWalks' default alpha = 1 - 1e-6 with tol = 1e-9 takes
    21 iterations on the 10 K-node graph of configs[0]) and results to 1e-6; dtype="float32" keeps the clamped f32 loop, dtype="float64"
    forces f64 at any tolerance."""
    import os
    import cases
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz"))
    eps32 = float(np.finfo(np.float32).eps)
    A, directed, p = cases.GRAPHS["er10k"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    pre = pg.preprocessor(assume_immutability=True)
    M = orc.normalize(A, "auto", directed)
    aw = pg.AbsorbingWalks(tol=1e-9, max_iters=1000, preprocessor=pre)
    got = np.asarray(aw.rank(graph, p.copy()).np, dtype=np.float64)
    assert aw.convergence.iteration == int(gold["er10k/absorbing_default|iters"]) == 21
    assert np.max(np.abs(got - gold["er10k/absorbing_default|ranks"])) <= 1e-6 * np.max(np.abs(gold["er10k/absorbing_default|ranks"]))
    clamped = pg.AbsorbingWalks(tol=1e-9, max_iters=1000, preprocessor=pre, dtype="float32")
    clamped.rank(graph, p.copy())
    assert clamped.convergence.iteration == orc.absorbing_walks(M, p, tol=1e-9, max_iters=1000, eps=eps32)[1] < 21
    # dtype="float64" at an ordinary tolerance: the oracle's fp64 loop, iteration for iteration
    for make, ref in ((lambda **k: pg.AbsorbingWalks(0.85, tol=1e-6, max_iters=1000, preprocessor=pre, **k),
                       lambda: orc.absorbing_walks(M, p, alpha=0.85, tol=1e-6, max_iters=1000)),
                      (lambda **k: pg.SymmetricAbsorbingRandomWalks(error_type=pg.L1, tol=1e-8, max_iters=1000, preprocessor=pre, **k),
                       lambda: orc.symmetric_absorbing_walks(M, p, error_type="l1", tol=1e-8, max_iters=1000)),
                      (lambda **k: pg.HeatKernel(3, tol=1e-10, max_iters=200, preprocessor=pre, **k),
                       lambda: orc.heat_kernel(M, p, t=3, tol=1e-10, max_iters=200)),
                      (lambda **k: pg.PageRankClosed(0.85, error_type=pg.L1, tol=1e-9, max_iters=1000, preprocessor=pre, **k),
                       lambda: orc.pagerank_closed(M, p, 0.85, error_type="l1", tol=1e-9, max_iters=1000))):
        want, want_iters = ref()
        for dtype in (None, "float64"):
            ranker = make(dtype=dtype)
            got = np.asarray(ranker.rank(graph, p.copy()).np, dtype=np.float64)
            if dtype is None and ranker.convergence.tol >= eps32:
                continue                                     # (an f32 run at an ordinary tolerance: covered by the golden cases)
            assert ranker.convergence.iteration == want_iters, (type(ranker).__name__, dtype, ranker.convergence.iteration, want_iters)
            assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), (type(ranker).__name__, dtype)
    with pytest.raises(Exception):
        pg.HeatKernel(3, dtype="float16")
    # a start ON the fixed point (a self-loop and an isolated node: p / |p| solves the walk's equation): the reference's fp64 loop sees
    # a zero residual at its first check and stops; the f64 walks form p / |p| in f64 themselves (an f32 quotient sums to 1 within 6e-8
    # only, and a run at 1e-9 started from it needs eight steps to come back)
    import scipy.sparse as sp
    loop = pg.AdjacencyWrapper(sp.csr_array(([2.0], ([0], [0])), shape=(2, 2)), directed=True)
    q = np.array([0.6, 1.3])
    Ml = orc.normalize(sp.csr_array(([2.0], ([0], [0])), shape=(2, 2)), "col", True)
    for make, ref in ((lambda: pg.AbsorbingWalks(0.85, error_type=pg.L1, tol=1e-9, max_iters=100, preprocessor=pg.preprocessor(normalization="col")),
                       lambda: orc.absorbing_walks(Ml, q, alpha=0.85, error_type="l1", tol=1e-9, max_iters=100)),
                      (lambda: pg.SymmetricAbsorbingRandomWalks(error_type=pg.L1, tol=1e-9, max_iters=100, preprocessor=pg.preprocessor(normalization="col")),
                       lambda: orc.symmetric_absorbing_walks(Ml, q, error_type="l1", tol=1e-9, max_iters=100))):
        ranker = make()
        got = np.asarray(ranker.rank(loop, q.copy()).np, dtype=np.float64)
        want, want_iters = ref()
        assert ranker.convergence.iteration == want_iters, (type(ranker).__name__, ranker.convergence.iteration, want_iters)
        assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), type(ranker).__name__


def check_differentiable_propagate(pg):
    """SURVEY.md 8f-4 / tests/test_gnn.py:22-28 (a ranker's propagate inside a model's forward pass): pygrank_amd.gnn.differentiable_propagate
    wraps the engine's multi-seed loop as a torch.autograd.Function whose backward pass is the same filter on the transposed operator.
    Forward against the oracle, the gradient against F^T applied by the oracle on the transposed matrix, on a directed ("col": M != M^T) and
    an undirected ("symmetric": M = M^T, the graph serves both passes) graph; a ranker that is not linear is refused."""
    import torch
    import cases
    from pygrank_amd.gnn import differentiable_propagate, transposed_operator
    rng = np.random.default_rng(5)
    for gkey in ("rmat10_dir", "er10k"):
        A, directed, _ = cases.GRAPHS[gkey]()
        n = A.shape[0]
        M = sp.csr_array(orc.normalize(A, "auto", directed))
        graph = pg.AdjacencyWrapper(A, directed=directed)
        pre = pg.preprocessor(assume_immutability=True)
        ranker = pg.PageRank(0.9, preprocessor=pre, use_quotient=False, error_type="iters", max_iters=10)      # tests/test_gnn.py:24-25
        X = torch.tensor(rng.random((n, 3)) * (rng.random((n, 3)) < 0.05), dtype=torch.float32, requires_grad=True)
        W = torch.tensor(rng.random((n, 3)), dtype=torch.float32)
        Y = differentiable_propagate(ranker, graph, X)
        loss = (Y * W).sum()
        loss.backward()
        Xn, Wn = X.detach().numpy().astype(np.float64), W.numpy().astype(np.float64)
        kw = dict(alpha=0.9, use_quotient=False, error_type="iters", max_iters=10)
        for j in range(3):
            want = orc.pagerank(M, Xn[:, j], **kw)[0] if Xn[:, j].any() else Xn[:, j]
            assert np.max(np.abs(Y.detach().numpy()[:, j] - want)) <= 2e-6 * max(np.max(np.abs(want)), 1e-30), (gkey, j)
            grad = orc.pagerank(sp.csr_array(M.T), Wn[:, j], **kw)[0]                  # F^T w: the same filter on the transposed matrix
            assert np.max(np.abs(X.grad.numpy()[:, j] - grad)) <= 2e-6 * np.max(np.abs(grad)), (gkey, j)
        same = transposed_operator(ranker, graph) is pre(graph)
        assert same == (not directed), gkey                                            # the symmetric operator serves both passes
    with pytest.raises(Exception):
        differentiable_propagate(pg.PageRank(0.9), graph, X)                           # the L1 quotient and a tolerance: not linear


def check_lazy_vectors_are_plain_vectors_to_every_observer(pg):
    """device.LazyVector (the backend-primitive route, pygrank/core/backend/__init__.py:59-80): whatever stays unevaluated -- conv,
    scalar products, a * conv + b * p, sums and differences in the engine's id space, |u - v| under sum / max -- has, for everybody who
    looks, the value the eager primitives give: random expression chains against numpy on the downloaded matrix, reductions, copies,
    item assignment, operands overwritten after the expression was formed, two graphs, and the fallbacks (vector products, powers,
    comparisons, scalar addition)."""
    import scipy.sparse as sp
    from pygrank_amd import device
    rng = np.random.default_rng(3)
    n = 200
    A = sp.csr_array(sp.random(n, n, density=0.05, random_state=5, format="csr"))
    A.data[:] = 1.0
    B = sp.csr_array(sp.random(n, n, density=0.08, random_state=6, format="csr"))
    pre = pg.preprocessor(normalization="col", assume_immutability=True)
    MA, MB = pre(pg.AdjacencyWrapper(A, directed=True)), pre(pg.AdjacencyWrapper(B, directed=True))
    TA, TB = [np.asarray(M.array.download_transposed().todense(), dtype=np.float64) for M in (MA, MB)]      # M^T, as stored
    x0, p0 = rng.random(n), rng.random(n)
    x, p = pg.to_array(x0), pg.to_array(p0)

    def close(got, want, tol=2e-6):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= tol * max(np.max(np.abs(want)), 1e-30), np.max(np.abs(got - want))

    c = pg.conv(x, MA)
    assert isinstance(c, device.LazyVector) and c._kind == "conv" and len(c) == n and c.shape == (n,)
    close(c, TA @ x0)
    assert c._kind == "res"                                         # looked at: evaluated once, in the id space and in the caller's ids
    close(pg.conv(x, MA) * 0.85 + p * 0.15, 0.85 * (TA @ x0) + 0.15 * p0)
    close(0.15 * p + 0.85 * pg.conv(x, MA), 0.85 * (TA @ x0) + 0.15 * p0)
    y = pg.conv(x, MA) * 0.85 + p * 0.15
    assert y._kind == "axpby"
    total = pg.sum(y)                                               # ONE engine step, sum(y) included
    assert y._kind == "res" and abs(total - (0.85 * (TA @ x0) + 0.15 * p0).sum()) <= 2e-6 * abs(total)
    q = y / total
    assert q._kind == "res" and q._res is y._res                    # a view of the same memory
    close(q, (0.85 * (TA @ x0) + 0.15 * p0) / total)
    want_y = (0.85 * (TA @ x0) + 0.15 * p0)
    z = pg.conv(q, MA)                                              # a resident operand: no way into the id space
    close(z, TA @ (want_y / total))
    d = z - q
    assert d._kind == "lin" and abs(d)._kind == "lin"
    want_d = TA @ (want_y / total) - want_y / total
    assert abs(pg.sum(pg.abs(z - q)) - np.abs(want_d).sum()) <= 2e-6 * np.abs(want_d).sum()
    assert abs(pg.max(pg.abs(z - q)) - np.abs(want_d).max()) <= 2e-6 * np.abs(want_d).max()
    assert abs(pg.Mabs(q)(z) - np.abs(want_d).mean()) <= 2e-6 * np.abs(want_d).mean()
    assert abs(pg.L1(z)(q) - np.abs(want_d).sum()) <= 2e-6 * np.abs(want_d).sum()
    assert abs(pg.MaxDifference(z)(q) - np.abs(want_d).max()) <= 2e-6 * np.abs(want_d).max()
    close(d, want_d)
    close(abs(z - q) * 2.0, 2.0 * np.abs(want_d))
    close(-(z - q), -want_d)
    close(z + q, TA @ (want_y / total) + want_y / total)
    close(z - p, TA @ (want_y / total) - p0)                        # a vector in the caller's ids joins the expression
    close(p - z, p0 - TA @ (want_y / total))
    # negative entries: max / min do not see the zero padding of the id space
    neg = pg.conv(x, MA) * -1.0
    assert abs(pg.max(neg) - (-(TA @ x0)).max()) <= 1e-6 and abs(pg.min(neg) - (-(TA @ x0)).min()) <= 1e-6
    assert abs(pg.sum(pg.abs(neg)) - np.abs(TA @ x0).sum()) <= 2e-6 * np.abs(TA @ x0).sum()
    # the fallbacks evaluate in the caller's ids
    close(pg.conv(x, MA) * p, (TA @ x0) * p0)
    close(pg.conv(x, MA) / (p + 1.0), (TA @ x0) / (p0 + 1.0))
    close(pg.conv(x, MA) + 1.0, (TA @ x0) + 1.0)
    close(pg.conv(x, MA) ** 2, (TA @ x0) ** 2)
    close(pg.conv(x, MA) > 0.1, ((TA @ x0) > 0.1).astype(np.float64))
    close(pg.exp(pg.conv(x, MA)), np.exp(TA @ x0), 1e-5)
    close(pg.conv(x, MA)[pg.to_array((p0 > 0.5).astype(np.float64))], (TA @ x0)[p0 > 0.5])
    assert abs(pg.dot(pg.conv(x, MA), p) - (TA @ x0) @ p0) <= 1e-5 * abs((TA @ x0) @ p0)
    assert abs(float(pg.conv(x, MA)[7]) - (TA @ x0)[7]) <= 1e-6
    # two graphs: an expression of one graph is an ordinary vector to the other
    close(pg.conv(pg.conv(x, MA), MB), TB @ (TA @ x0))
    close(pg.conv(x, MA) + pg.conv(x, MB), TA @ x0 + TB @ x0)
    close(pg.conv(x, MA) * 0.5 + pg.conv(p, MB) * 2.0, 0.5 * (TA @ x0) + 2.0 * (TB @ p0))
    # an operand overwritten AFTER the expression was formed: the expression keeps the old values
    w = pg.to_array(x0.copy())
    e1, e2, e3 = w * 3.0, pg.conv(w, MA), pg.conv(x, MA) * 0.5 + w * 2.0
    w[5] = 123.0
    close(e1, 3.0 * x0)
    close(e2, TA @ x0)
    close(e3, 0.5 * (TA @ x0) + 2.0 * x0)
    x5 = x0.copy()
    x5[5] = 123.0
    close(w, x5)
    close(pg.conv(w, MA), TA @ x5)                                   # ... and the remembered resident copy of w went with the write
    # item assignment into an expression: it becomes memory of its own
    e = pg.conv(x, MA) * 2.0
    e[3] = -1.0
    want_e = 2.0 * (TA @ x0)
    want_e[3] = -1.0
    close(e, want_e)
    close(e * 1.0 + pg.conv(x, MA), want_e + TA @ x0)
    k = pg.copy(pg.conv(x, MA))
    close(k, TA @ x0)
    # a closed-form filter keeps the product as its next power: evaluated once, shared
    term = pg.conv(x, MA)
    result = p + term * 0.3
    nxt = pg.conv(term, MA)
    close(result, p0 + 0.3 * (TA @ x0))
    close(nxt, TA @ (TA @ x0))
    # the absorbing walk's formula piece by piece (adhoc.py:166-169), whole and interrupted at every stage
    deg, lam = pg.to_array(rng.random(n) + 0.5), pg.to_array(rng.random(n) + 0.1)
    deg0, lam0 = np.asarray(deg), np.asarray(lam)
    walk = (pg.conv(x, MA) * deg + p * lam) / (lam + deg)
    assert walk._kind == "walk"
    close(walk, ((TA @ x0) * deg0 + p0 * lam0) / (lam0 + deg0))
    assert abs(pg.sum((pg.conv(x, MA) * 0.5 * deg + p * lam) / (lam + deg)) - ((0.5 * (TA @ x0) * deg0 + p0 * lam0) / (lam0 + deg0)).sum()) <= 1e-4
    close(pg.conv(x, MA) * deg, (TA @ x0) * deg0)                                       # "cmul", looked at
    close(pg.conv(x, MA) * deg + p * lam, (TA @ x0) * deg0 + p0 * lam0)                 # "cmul_add", looked at
    close((pg.conv(x, MA) * deg + p * lam) / (lam + p), ((TA @ x0) * deg0 + p0 * lam0) / (lam0 + p0))      # not the walk: another divisor
    close((pg.conv(x, MA) * deg + p * lam) / 2.0, ((TA @ x0) * deg0 + p0 * lam0) / 2.0)
    close(p * lam, p0 * lam0)                                                           # "vv": an ordinary product to whoever looks
    close((p * lam) * 2.0 + (lam + deg), 2.0 * p0 * lam0 + lam0 + deg0)
    close(deg * pg.conv(x, MA), deg0 * (TA @ x0))
    held = p * lam
    p[2] = 9.0                                                                          # written after the product was formed
    close(held, p0 * lam0)
    p[2] = float(p0[2])
    # a filter that never looks at its iterate nests one expression per step: the chain is evaluated before it can exhaust the stack
    it, want_it = x, x0
    for _ in range(1500):
        it = pg.conv(it, MA) * 0.5 + p * 0.5
        want_it = 0.5 * (TA @ want_it) + 0.5 * p0
    close(it, want_it, 1e-5)
    # the switch: every primitive evaluated where it stands
    device.LAZY = False
    try:
        eager = pg.conv(x, MA) * 0.85 + p * 0.15
        assert type(eager) is device.DeviceVector
        close(eager, want_y)
    finally:
        device.LAZY = True


def check_backend_primitive_route_is_one_step_per_formula(pg):
    """The route the unmodified reference filters take (one backend primitive at a time): per PageRank iteration ONE engine step
    (a * M^T x + b * p with sum(y)) and ONE residual, no pgh_spmv, nothing in the caller's ids until the result is looked at; a
    closed-form filter: one step + one pgh_axpby per term.  Counted at the ctypes boundary."""
    import collections
    import cases
    from pygrank_amd import _lib as L
    A, directed, p = cases.GRAPHS["rmat10_dir"]()
    graph = pg.AdjacencyWrapper(A, directed=directed)
    pre = pg.preprocessor(assume_immutability=True)
    lib = L.lib()
    counts = collections.Counter()
    watched = ("pgh_spmv", "pgh_resident_step", "pgh_resident_in", "pgh_resident_out", "pgh_resident_gather", "pgh_scaled_residual",
               "pgh_residual", "pgh_axpby", "pgh_ewise_vv", "pgh_ewise_vs", "pgh_reduce")
    originals = {name: getattr(lib, name) for name in watched}

    def counted(name):
        fn = originals[name]

        def call(*args):
            counts[name] += 1
            return fn(*args)
        return call
    try:
        for name in watched:
            setattr(lib, name, counted(name))
        for make, per_iteration in ((lambda: pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=1000),
                                     dict(pgh_resident_step=1, pgh_scaled_residual=1)),
                                    (lambda: pg.HeatKernel(5, preprocessor=pre, error_type="iters", max_iters=12), dict(pgh_resident_step=1, pgh_axpby=1)),
                                    # AbsorbingWalks._formula (adhoc.py:166-169) recognised whole: conv * deg, + p * lam, / (lam + deg)
                                    (lambda: pg.AbsorbingWalks(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=1000),
                                     dict(pgh_resident_step=1, pgh_scaled_residual=1))):
            ranker = make()
            ranker._fused_loop = lambda *a, **k: False
            ranker._fused_rank = lambda *a, **k: None
            sig = pg.to_signal(graph, p)
            ranker.rank(graph, sig)                      # (the first run also uploads the graph)
            counts.clear()
            out = ranker.rank(graph, sig)
            steps = ranker.convergence.iteration - 1
            assert steps >= 8
            assert counts["pgh_spmv"] == 0 and counts["pgh_resident_out"] == 0, dict(counts)      # nobody has looked yet
            np.asarray(out.np)                           # (a filter that never looks at its iterate is evaluated here, or chain by chain)
            assert counts["pgh_resident_out"] == 1
            for name, each in per_iteration.items():
                assert steps - 2 <= counts[name] <= steps * each + 1, (name, counts[name], steps)
            assert counts["pgh_resident_in"] <= 4 and counts["pgh_ewise_vv"] <= 2 and counts["pgh_resident_gather"] <= 1, dict(counts)
            fused = make()
            want = np.asarray(fused.rank(graph, sig).np)
            assert fused.convergence.iteration == ranker.convergence.iteration
            assert np.max(np.abs(np.asarray(out.np) - want)) <= 1e-6 * np.max(np.abs(want))
    finally:
        for name, fn in originals.items():
            setattr(lib, name, fn)


def check_algorithms_describe_themselves(pg):
    """NodeRanking.cite / references / __str__ (signals.py:228-249; abstract_filters.py:79-86,145-149,186-194): a list of parts,
    read out as "first with second, third and last"; a chained personalization transform is cited ahead of the filter."""
    plain = pg.PageRank(0.85)
    assert plain.references() == [plain._reference()] and str(plain) == plain.cite() == plain._reference()
    eig = pg.PageRank(0.9, converge_to_eigenvectors=True)
    assert len(eig.references()) == 2 and " with " in eig.cite() and " and " not in eig.cite()
    cheb = pg.HeatKernel(3, coefficient_type="chebyshev", optimization_dict=dict())
    refs = cheb.references()
    assert len(refs) == 3 and cheb.cite() == refs[0] + " with " + refs[1] + " and " + refs[2]
    post = pg.Normalize(pg.Ordinals(pg.AbsorbingWalks(0.9)))
    assert len(post.references()) == 3 and post.references()[0] == pg.AbsorbingWalks(0.9)._reference()
    chained = pg.PageRank(0.5) >> pg.HeatKernel(2)
    assert chained.cite().endswith("passed to " + pg.HeatKernel(2).cite()) and chained.cite().startswith(pg.PageRank(0.5).cite())
    for algo in (pg.SymmetricAbsorbingRandomWalks(), pg.GenericGraphFilter([0.5, 0.5]), pg.PageRankClosed(0.7), pg.Top(3), pg.Threshold(0.1),
                 pg.Sweep(pg.PageRank()), pg.LinearSweep(pg.PageRank()), pg.Transformer(), pg.Tautology()):
        assert isinstance(str(algo), str) and len(str(algo)) > 3


ALL = [v for k, v in sorted(globals().items()) if k.startswith("check_") and callable(v)]
