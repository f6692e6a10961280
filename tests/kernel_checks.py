"""Engine-agnostic kernel-level checks through the C-ABI wrappers (DeviceVector / DeviceGraph / fused steps).
Run against the host test double on the CPU (validates the checks themselves) and against libpgh_hip.so on
the MI355X (tests/test_gpu_parity.py).  Tolerances are written next to each assertion."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from oracle import ref_loops as orc, rmat_np

F32 = np.float32
EPS32 = float(np.finfo(np.float32).eps)


def _vec(pg, a):
    return pg.to_array(np.asarray(a, dtype=np.float64))


def _np(v):
    return np.asarray(v, dtype=np.float64)


SIZES = [0, 1, 3, 64, 255, 256, 1000, 4099, 100003]


def check_elementwise_and_reductions(pg):
    rng = np.random.default_rng(0)
    for n in SIZES:
        a = (rng.random(n) * 4 - 2).astype(F32).astype(np.float64)
        b = (rng.random(n) * 3 + 0.5).astype(F32).astype(np.float64)
        da, db = _vec(pg, a), _vec(pg, b)
        f = lambda x: np.asarray(x, dtype=F32)               # noqa: E731  (engine arithmetic is fp32)
        # single-rounding ops are bit-exact against numpy fp32
        for k, (got, want) in enumerate([(da + db, f(a) + f(b)), (da - db, f(a) - f(b)), (da * db, f(a) * f(b)),
                                         (da * 2.5, f(a) * F32(2.5)), (3.0 - da, F32(3.0) - f(a)), (-da, -f(a)),
                                         (abs(da), np.abs(f(a))), (da > db, (f(a) > f(b)).astype(F32))]):
            assert np.array_equal(np.asarray(got, dtype=F32), want.astype(F32)), (n, k)
        # division and the fused a*x + b*y (hipcc contracts it into an FMA) are held to 1 ulp
        for k, (got, want) in enumerate([(da / db, a / b), (2.0 / db, 2.0 / b)]):
            assert np.allclose(_np(got), want, rtol=1.5 * EPS32, atol=1e-30), (n, k)
        want = float(F32(0.3)) * a + float(F32(0.7)) * b
        assert np.all(np.abs(_np(da.axpby(0.3, db, 0.7)) - want) <= 2 * EPS32 * (0.3 * np.abs(a) + 0.7 * np.abs(b)) + 1e-30), n
        assert np.allclose(_np(pg.exp(da)), np.exp(a), rtol=1e-6)
        assert np.allclose(_np(pg.log(db)), np.log(b), rtol=1e-6, atol=1e-6)
        assert np.allclose(_np(db ** 2), b ** 2, rtol=1e-6)
        # f64-accumulated reductions of the f32 data: exact to 1e-12 relative
        assert abs(pg.sum(da) - a.sum()) <= 1e-12 * max(1.0, np.abs(a).sum())
        assert abs(pg.sum(pg.abs(da)) - np.abs(a).sum()) <= 1e-12 * max(1.0, np.abs(a).sum())
        assert abs(pg.dot(da, db) - float(a @ b)) <= 1e-12 * max(1.0, float(np.abs(a) @ np.abs(b)))
        if n:
            assert pg.max(da) == a.max() and pg.min(da) == a.min()
            assert abs(pg.Mabs(da)(db) - np.abs(a - b).sum() / n) <= 1e-12 * max(1.0, np.abs(a - b).sum())
            assert abs(pg.L1(da)(db) - np.abs(a - b).sum()) <= 1e-12 * max(1.0, np.abs(a - b).sum())
            assert pg.MaxDifference(da)(db) == np.abs(a - b).max()
            keep = rng.random(n) < 0.5
            got = pg.filter_out(da, _vec(pg, (~keep).astype(float)))
            assert np.array_equal(_np(got), a[keep])
            assert np.array_equal(_np(da.copy()), a) and float(da[n - 1]) == a[-1]


def _random_matrix(rng, n_rows, n_cols, density, hubs=0, empty_frac=0.0):
    A = sp.random(n_rows, n_cols, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))),
                  format="lil")
    for _ in range(hubs):
        A[:, int(rng.integers(0, n_cols))] = rng.random((n_rows, 1))      # hub column of M = hub row of M^T
    A = sp.csr_array(A.tocsr())
    if empty_frac:
        keep = (rng.random(n_cols) >= empty_frac).astype(float)
        A = sp.csr_array(A @ sp.diags(keep))
        A.eliminate_zeros()
    A.sort_indices()
    return A


def matrices():
    rng = np.random.default_rng(1)
    yield "empty5", sp.csr_array((5, 5))
    yield "single", sp.csr_array(np.array([[2.0]]))
    yield "dense40", sp.csr_array(rng.random((40, 40)))
    yield "rect_30x70", _random_matrix(rng, 30, 70, 0.2)
    yield "rect_70x30", _random_matrix(rng, 70, 30, 0.2)
    yield "sparse_3000", _random_matrix(rng, 3000, 3000, 0.002, hubs=2, empty_frac=0.3)
    yield "hubs_5000", _random_matrix(rng, 5000, 5000, 0.001, hubs=5, empty_frac=0.6)
    yield "empty_tail", sp.csr_array(sp.vstack([_random_matrix(rng, 100, 20000, 0.05), sp.csr_array((19900, 20000))]).T)
    yield "rmat14", rmat_np.rmat_csr(14, 8, seed=2)
    yield "rmat16_ef16", rmat_np.rmat_csr(16, 16, seed=0)


def check_upload_transpose_degrees_spmv(pg):
    rng = np.random.default_rng(2)
    for name, M in matrices():
        g = pg.scipy_sparse_to_backend(M)
        assert g.shape == M.shape
        MT = g.download_transposed()
        want = sp.csr_array(M.T.astype(F32))
        want.sort_indices()
        # bit-exact format conversion: same structure, values rounded once to f32, columns ascending per row
        assert np.array_equal(MT.indptr, want.indptr), name
        assert np.array_equal(MT.indices, want.indices), name
        assert np.array_equal(MT.data, want.data), name
        deg = np.asarray(M.sum(axis=1)).ravel()
        assert np.allclose(_np(pg.degrees(g)), deg, rtol=EPS32, atol=0), name
        x = rng.random(M.shape[0]).astype(F32).astype(np.float64)
        y = _np(pg.conv(_vec(pg, x), g))
        ref = x @ sp.csr_array(M.astype(F32).astype(np.float64))     # exact sum of the engine's f32 products, up to
        scale = np.abs(x) @ np.abs(M)                                   # one f32 rounding per product and per row
        assert np.all(np.abs(y - ref) <= 2.5 * EPS32 * scale + 1e-30), name


def check_upload_rejects_invalid_csr(pg):
    """scipy (what the reference hands its matrices to) refuses a CSR whose column indices leave [0, n_cols) or whose row
    pointers decrease; so does the engine, instead of writing outside its row-pointer array."""
    import ctypes as C
    from pygrank_amd import _lib as L
    L.ensure_init()
    lib = L.lib()

    def upload(indptr, indices, data, n_rows, n_cols):
        ip, ix, dt = np.asarray(indptr, np.int64), np.asarray(indices, np.int32), np.asarray(data, np.float64)
        h = L.c_graph()
        rc = lib.pgh_graph_from_csr(n_rows, n_cols, len(ix), ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p),
                                    dt.ctypes.data_as(C.c_void_p), 0, C.byref(h))
        if rc == 0:
            lib.pgh_graph_destroy(h)
        return rc, lib.pgh_last_error()
    assert upload([0, 2, 3], [0, 1, 2], [1., 1., 1.], 2, 3)[0] == 0
    rc, msg = upload([0, 2, 3], [0, 3, 2], [1., 1., 1.], 2, 3)          # index == n_cols
    assert rc != 0 and b"column index" in msg
    rc, msg = upload([0, 2, 3], [0, -1, 2], [1., 1., 1.], 2, 3)         # negative index
    assert rc != 0 and b"column index" in msg
    rc, msg = upload([0, 3, 2, 3], [0, 1, 2], [1., 1., 1.], 3, 3)       # decreasing row pointers
    assert rc != 0 and b"indptr" in msg


def check_graph_dropout(pg):
    """graph_dropout(M, rate) with rate > 0 (specification.py:13; pytorch.py:34-38): the mask is a hash of
    (seed, entry of CSR(M^T)), so the product can be checked EXACTLY against scipy on the masked matrix rebuilt with the
    numpy twin of the hash (oracle/rmat_np.splitmix64); plus the statistics of a dropout mask and the filter-level hooks."""
    A = rmat_np.rmat_csr(11, 8, seed=2)
    M = sp.csr_array(orc.normalize(A, "col", True))
    g = pg.scipy_sparse_to_backend(M)
    MT = g.download_transposed()                                   # entry order of the mask
    rng = np.random.default_rng(8)
    x = rng.random(M.shape[0]).astype(F32).astype(np.float64)
    assert pg.graph_dropout(g, 0) is g                             # identity and O(1)
    for rate in (0.1, 0.5, 0.9):
        pg.backend.hip.set_dropout_seed(41)
        dropped = pg.graph_dropout(g, rate)
        e = np.arange(MT.nnz, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = rmat_np.splitmix64(np.uint64(dropped.seed) ^ (e * np.uint64(0xD6E8FEB86659FD93)))
        keep = (h >> np.uint64(32)).astype(np.int64) >= int(np.floor(rate * 4294967296.0))
        assert abs(keep.mean() - (1 - rate)) < 0.02                # Bernoulli(1 - rate) survivors
        masked = sp.csr_array(((MT.data * np.float32(1.0 / (1.0 - rate))).astype(F32).astype(np.float64) * keep, MT.indices, MT.indptr), shape=MT.shape)
        got = _np(pg.conv(_vec(pg, x), dropped))
        ref = masked @ x
        scale = np.abs(masked) @ np.abs(x)
        assert np.all(np.abs(got - ref) <= 2.5 * EPS32 * scale + 1e-30), rate
        # degrees of the dropped graph = row sums of the masked M (pytorch.py:100-104 on the dropped matrix): column sums of
        # the masked CSR(M^T), exactly rebuilt here
        deg = _np(pg.degrees(dropped))
        deg_ref = np.asarray(masked.sum(axis=0)).ravel()
        assert np.all(np.abs(deg - deg_ref) <= 4 * EPS32 * np.abs(deg_ref) + 1e-30), rate
        again = pg.graph_dropout(g, rate)                          # a new mask at every call (abstract_filters.py:59-62)
        assert again.seed != dropped.seed and not np.array_equal(_np(pg.conv(_vec(pg, x), again)), got)
    # E[dropout] = identity: the mean over masks approaches the plain product
    plain = _np(pg.conv(_vec(pg, x), g))
    acc = np.zeros_like(plain)
    for _ in range(200):
        acc += _np(pg.conv(_vec(pg, x), pg.graph_dropout(g, 0.5)))
    assert np.abs(acc / 200 - plain).sum() <= 0.12 * np.abs(plain).sum()
    # filter level: rank(..., graph_dropout=) is ONE device loop (pgh_ppr_run_dropout) with a fresh mask per step inside the step's
    # kernels -- against a host loop that rebuilds every step's mask, against the hook protocol (one engine call per backend
    # primitive: fused_dropout = False) from the same seed, and deterministic under a fixed seed
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(normalization="col", assume_immutability=True)
    p = np.zeros(A.shape[0])
    p[:20] = 1.0
    outs = []
    for _ in range(2):
        pg.backend.hip.set_dropout_seed(7)
        ranker = pg.PageRank(0.85, preprocessor=pre, error_type="iters", max_iters=12)
        outs.append(np.asarray(ranker.rank(graph, p.copy(), graph_dropout=0.3).np))
        assert ranker.last_loop["spmv"] == 11 and ranker.convergence.iteration == 12
    assert np.array_equal(outs[0], outs[1]) and abs(outs[0].sum() - 20.0) < 1e-3
    gm = getattr(pre(graph), "array", pre(graph))
    MTm = gm.download_transposed()

    def masked_at(seed, rate=0.3):
        e = np.arange(MTm.nnz, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = rmat_np.splitmix64(np.uint64(seed) ^ (e * np.uint64(0xD6E8FEB86659FD93)))
        keep = (h >> np.uint64(32)).astype(np.int64) >= int(np.floor(rate * 4294967296.0))
        data = (MTm.data.astype(F32) * np.float32(1.0 / (1.0 - rate))).astype(F32).astype(np.float64) * keep
        return sp.csr_array((data, MTm.indices, MTm.indptr), shape=MTm.shape)
    pn = (p.astype(F32) / np.float32(20.0)).astype(np.float64)
    x, quot = pn.copy(), 1.0
    for k in range(11):                                            # seed 7: _start draws mask 8, step k + 1 runs on mask 9 + k
        y = 0.85 * quot * (masked_at(9 + k) @ x) + 0.15 * pn
        quot, x = 1.0 / y.sum(), y
    want = x * quot * 20.0
    assert np.max(np.abs(outs[0] - want)) <= 2e-6 * np.max(np.abs(want))
    pg.backend.hip.set_dropout_seed(7)
    hooks = pg.PageRank(0.85, preprocessor=pre, error_type="iters", max_iters=12)
    hooks.fused_dropout = False
    by_hooks = np.asarray(hooks.rank(graph, p.copy(), graph_dropout=0.3).np)
    assert not hasattr(hooks, "last_loop") and hooks.convergence.iteration == 12
    assert np.max(np.abs(by_hooks - outs[0])) <= 2e-6 * np.max(np.abs(outs[0]))
    # both routes leave the seed counter in the same place (iterations + 1 masks drawn: _start, every step, _end), so consecutive
    # rank(..., graph_dropout=) calls after ONE set_dropout_seed are reproducible across the routes ...
    assert pg.backend.hip.peek_dropout_seed() == 7 + 13 + 1
    second = []
    for fused in (True, False):
        pg.backend.hip.set_dropout_seed(7)
        both = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=0.35, max_iters=200)     # ... also when a tolerance stops the run early
        both.fused_dropout = fused
        both.rank(graph, p.copy(), graph_dropout=0.3)
        assert pg.backend.hip.peek_dropout_seed() == 7 + both.convergence.iteration + 1 + 1, (fused, both.convergence.iteration)
        second.append(np.asarray(both.rank(graph, p.copy(), graph_dropout=0.3).np))
    assert np.max(np.abs(second[0] - second[1])) <= 2e-6 * np.max(np.abs(second[1]))
    # a tolerance instead of a count: the loop stops on the device, on the residual of the masked iteration
    pg.backend.hip.set_dropout_seed(7)
    stopping = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=0.35, max_iters=200)
    by_tol = np.asarray(stopping.rank(graph, p.copy(), graph_dropout=0.3).np)
    x, quot, prev, prev_quot, its = pn.copy(), 1.0, None, 1.0, 1
    while its < 200:
        if prev is not None and np.abs(x * quot - prev * prev_quot).sum() <= 0.35:
            break
        y = 0.85 * quot * (masked_at(9 + its - 1) @ x) + 0.15 * pn
        prev, prev_quot, x, quot, its = x, quot, y, 1.0 / y.sum(), its + 1
    assert 2 < its < 200 and stopping.convergence.iteration == its
    assert np.max(np.abs(by_tol - x * quot * 20.0)) <= 2e-6 * np.max(np.abs(x * quot)) * 20.0
    base = np.asarray(pg.PageRank(0.85, preprocessor=pre, error_type="iters", max_iters=12).rank(graph, p.copy()).np)
    assert not np.allclose(outs[0], base) and np.corrcoef(outs[0], base)[0, 1] > 0.9
    # filters that ask for degrees(M) in _start run with a dropped graph too
    for algo in (pg.AbsorbingWalks(0.85, error_type="iters", max_iters=8), pg.SymmetricAbsorbingRandomWalks(error_type="iters", max_iters=8)):
        pg.backend.hip.set_dropout_seed(11)
        out = np.asarray(algo.rank(graph, p.copy(), graph_dropout=0.3).np)
        assert np.all(np.isfinite(out)) and out.sum() > 0


def dropout_on_cold_image_check(pg):
    """graph_dropout on a graph whose cold entries live in the propagation-blocking image (scale 18: run with PGH_PB=1 PGH_PB_FORCE=1
    PGH_BLOCKS=4, tests/test_gpu_parity.py does): the stream kernel AND phase A multiply by the mask of the entry's index in CSR(M^T)
    order -- conv against scipy on the rebuilt mask (multigraph: repeated entries share one bit), the device loop against a host loop."""
    A = rmat_np.rmat_csr(18, 8, seed=3)
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(normalization="col", assume_immutability=True)
    g = getattr(pre(graph), "array", pre(graph))
    assert "propagation-blocking image" in g.format(), g.format()
    MT = g.download_transposed()
    rng = np.random.default_rng(12)
    x = rng.random(A.shape[0]).astype(F32).astype(np.float64)

    def masked_at(seed, rate):
        e = np.arange(MT.nnz, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = rmat_np.splitmix64(np.uint64(seed) ^ (e * np.uint64(0xD6E8FEB86659FD93)))
        keep = (h >> np.uint64(32)).astype(np.int64) >= int(np.floor(rate * 4294967296.0))
        data = (MT.data.astype(F32) * np.float32(1.0 / (1.0 - rate))).astype(F32).astype(np.float64) * keep
        return sp.csr_array((data, MT.indices, MT.indptr), shape=MT.shape)
    for rate in (0.2, 0.7):
        pg.backend.hip.set_dropout_seed(60)
        dropped = pg.graph_dropout(g, rate)
        got = _np(pg.conv(_vec(pg, x), dropped))
        masked = masked_at(dropped.seed, rate)
        ref, scale = masked @ x, np.abs(masked) @ np.abs(x)
        assert np.all(np.abs(got - ref) <= 4 * EPS32 * scale + 1e-30), rate
    p = np.zeros(A.shape[0])
    p[rng.choice(A.shape[0], 50, replace=False)] = 1.0
    pg.backend.hip.set_dropout_seed(200)
    ranker = pg.PageRank(0.85, preprocessor=pre, error_type="iters", max_iters=8)
    out = np.asarray(ranker.rank(graph, p.copy(), graph_dropout=0.3).np)
    assert ranker.last_loop["spmv"] == 7
    pn = (p.astype(F32) / np.float32(50.0)).astype(np.float64)
    xk, quot = pn.copy(), 1.0
    for k in range(7):                                             # seed 200: _start draws mask 201, step k + 1 runs on mask 202 + k
        y = 0.85 * quot * (masked_at(202 + k, 0.3) @ xk) + 0.15 * pn
        quot, xk = 1.0 / y.sum(), y
    want = xk * quot * 50.0
    assert np.max(np.abs(out - want)) <= 2e-6 * np.max(np.abs(want))


def check_graph_dropout_batched(pg):
    """graph_dropout inside the multi-seed kernel ("per-iteration edge masking in the SpMM kernel"): a slab
    product against scipy on the rebuilt mask, and propagate(..., graph_dropout=) as ONE batched device loop against a host
    loop that rebuilds the mask of every step (seed0 + k - 1) -- on a multigraph (value-free stream, repeated entries share
    one mask bit) and on a weighted graph (valued stream)."""
    def mask_of(MT, rate, seed):
        e = np.arange(MT.nnz, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = rmat_np.splitmix64(np.uint64(seed) ^ (e * np.uint64(0xD6E8FEB86659FD93)))
        keep = (h >> np.uint64(32)).astype(np.int64) >= int(np.floor(rate * 4294967296.0))
        data = (MT.data.astype(F32) * np.float32(1.0 / (1.0 - rate))).astype(F32).astype(np.float64) * keep
        return sp.csr_array((data, MT.indices, MT.indptr), shape=MT.shape)

    rng = np.random.default_rng(21)
    A = rmat_np.rmat_csr(11, 8, seed=5)                            # duplicate edges: weights 1, 2, 3 ...
    W = sp.csr_array(A)
    W.data = rng.uniform(0.5, 2.0, W.nnz)
    for label, adj in (("multigraph", A), ("weighted", W)):
        n = adj.shape[0]
        graph = pg.AdjacencyWrapper(adj, directed=True)
        pre = pg.preprocessor(normalization="col", assume_immutability=True)
        g = pre(graph)
        g = getattr(g, "array", g)
        MT = g.download_transposed()
        rate, b = 0.3, 5
        X = rng.random((n, b)).astype(F32).astype(np.float64)
        pg.backend.hip.set_dropout_seed(100)
        dropped = pg.graph_dropout(g, rate)
        got = np.asarray(pg.conv(pg.to_primitive(X), dropped))
        masked = mask_of(MT, rate, dropped.seed)
        ref, scale = masked @ X, np.abs(masked) @ np.abs(X)
        assert got.shape == ref.shape and np.all(np.abs(got - ref) <= 2.5 * EPS32 * scale + 1e-30), label
        # ---- the batched loop: 7 iterations = 6 steps, each with its own mask
        F = np.zeros((n, b))
        for j in range(b):
            F[rng.choice(n, 15, replace=False), j] = rng.random(15) + 0.5
        ranker = pg.PageRank(0.85, preprocessor=pre, error_type="iters", max_iters=7)
        pg.backend.hip.set_dropout_seed(500)
        out = np.asarray(ranker.propagate(graph, pg.to_primitive(F), graph_dropout=rate))
        assert hasattr(ranker, "last_batches") and all(c["spmv"] == 6 for c in ranker.last_batches[0]), label
        for j in range(b):
            norm = np.abs(F[:, j]).sum()
            pj = (F[:, j].astype(F32) / np.float32(norm)).astype(np.float64)
            x, quot = pj.copy(), 1.0
            for k in range(6):
                y = 0.85 * quot * (mask_of(MT, rate, 501 + k) @ x) + 0.15 * pj
                quot, x = 1.0 / y.sum(), y
            want = x * quot * norm
            assert np.max(np.abs(out[:, j] - want)) <= 2e-6 * np.max(np.abs(want)), (label, j)
        # a different seed gives a different outcome; rate 0 through the same entry point is the plain batch
        pg.backend.hip.set_dropout_seed(900)
        other = np.asarray(ranker.propagate(graph, pg.to_primitive(F), graph_dropout=rate))
        assert not np.array_equal(other, out)
        plain = np.asarray(ranker.propagate(graph, pg.to_primitive(F)))
        zero = np.asarray(ranker.propagate(graph, pg.to_primitive(F), graph_dropout=0))
        assert np.array_equal(plain, zero)


def check_fused_steps(pg):
    from pygrank_amd import _lib as L
    from pygrank_amd.device import DeviceVector
    rng = np.random.default_rng(3)
    for name, M in matrices():
        if M.shape[0] != M.shape[1]:
            continue
        n = M.shape[0]
        g = pg.scipy_sparse_to_backend(M)
        M32 = sp.csr_array(M.astype(F32).astype(np.float64))
        x = rng.random(n).astype(F32).astype(np.float64)
        p = rng.random(n).astype(F32).astype(np.float64)
        dx, dp = _vec(pg, x), _vec(pg, p)
        tol = lambda ref_abs: 4 * EPS32 * ref_abs + 1e-30           # noqa: E731
        # ---- PageRank step (adhoc.py:36) with a pending quotient
        y = DeviceVector.empty(n)
        s = C.c_double()
        L.check(L.lib().pgh_ppr_step(g._h, dx._h, 0.5, dp._h, 0.85, y._h, C.byref(s)))
        ref = 0.85 * 0.5 * (x @ M32) + 0.15 * p
        bound = 0.85 * 0.5 * (np.abs(x) @ np.abs(M32)) + 0.15 * np.abs(p)
        assert np.all(np.abs(_np(y) - ref) <= tol(bound)), name
        assert abs(s.value - _np(y).sum()) <= 1e-12 * max(1.0, np.abs(_np(y)).sum()), name
        # ---- AbsorbingWalks step (adhoc.py:167-168)
        deg = (rng.random(n) + 0.1).astype(F32).astype(np.float64)
        lam = (rng.random(n) + 0.1).astype(F32).astype(np.float64)
        ddeg, dlam = _vec(pg, deg), _vec(pg, lam)
        L.check(L.lib().pgh_absorb_step(g._h, dx._h, 2.0, dp._h, ddeg._h, dlam._h, y._h, C.byref(s)))
        ref = ((2.0 * (x @ M32)) * deg + p * lam) / (lam + deg)
        bound = ((2.0 * (np.abs(x) @ np.abs(M32))) * deg + p * lam) / (lam + deg)
        assert np.all(np.abs(_np(y) - ref) <= 2 * tol(bound)), name
        assert abs(s.value - _np(y).sum()) <= 1e-12 * max(1.0, np.abs(_np(y)).sum()), name
        # ---- polynomial step (abstract_filters.py:215-230): taylor and the chebyshev recurrence
        for a, b in ((1.0, 0.0), (2.0, -1.0)):
            res0 = rng.random(n).astype(F32).astype(np.float64)
            dres, tout, d = _vec(pg, res0), DeviceVector.empty(n), C.c_double()
            L.check(L.lib().pgh_poly_step(g._h, dx._h, tout._h, a, b, dres._h, 0.25, L.ERR_L1, C.byref(d)))
            t_ref = a * (x @ M32) + b * x
            t_bound = abs(a) * (np.abs(x) @ np.abs(M32)) + abs(b) * np.abs(x)
            assert np.all(np.abs(_np(tout) - t_ref) <= tol(t_bound)), name
            r_ref = res0 + 0.25 * _np(tout)
            assert np.all(np.abs(_np(dres) - r_ref) <= 2 * EPS32 * np.abs(r_ref) + 1e-30), name
            assert abs(d.value - np.abs(_np(dres) - res0).sum()) <= 1e-9 * max(1.0, np.abs(r_ref).sum()), name
        # ---- scaled residual (abstract_filters.py:133-134 + supervised.py:93-138)
        for kind, fn in ((L.ERR_L1, lambda v: v.sum()), (L.ERR_MABS, lambda v: v.sum() / n), (L.ERR_LINF, lambda v: v.max())):
            e = C.c_double()
            L.check(L.lib().pgh_scaled_residual(kind, dx._h, 0.7, dp._h, 1.3, C.byref(e)))
            want = fn(np.abs(x * 0.7 - p * 1.3))
            assert abs(e.value - want) <= 1e-12 * max(1.0, want), name


def check_chebyshev_f64_route_on_blocked_stream(pg):
    """The f64 route of the "chebyshev" recurrence (abstract_filters.py:216-224) on graphs large enough for its own blocked
    image (pgh_bsf64.hip: 8 / 16 / 32 column blocks, LDS hot cache + in-stream cold gathers, value-free and valued streams)
    against the oracle at 1e-6 with equal iteration counts, and against the row-major CSR route of the same engine."""
    import os
    from oracle import ref_loops as orc
    rng = np.random.default_rng(11)
    A = rmat_np.rmat_csr(15, 16, seed=4)                       # 32 768 nodes: beyond one hot cache (20 224 doubles)
    W = sp.csr_array(A)
    W.data = rng.uniform(0.5, 2.0, W.nnz)                      # weights that do not factor: the valued stream
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 40, seed=2)] = rng.uniform(0.5, 1.5, 40)
    saved = {k: os.environ.get(k) for k in ("PGH_BLOCKS64", "PGH_CHEB_CSR")}
    try:
        for label, graph, normalization, blocks in (("col", A, "col", None), ("symmetric", sp.csr_array(A + A.T), "symmetric", None),
                                                    ("weighted", W, "col", None), ("col/16", A, "col", "16"), ("col/32", A, "col", "32")):
            M = orc.normalize(graph, normalization, True)
            # (round 6: a tolerance below fp32 eps is honoured as the reference's fp64 engine honours it -- the route is f64 anyway)
            want, want_iters = orc.heat_kernel(M, p, t=5, coefficient_type="chebyshev", tol=1e-9, max_iters=40, error_type="l1")
            results = {}
            for route in ("blocked", "csr"):
                os.environ.pop("PGH_BLOCKS64", None)
                os.environ.pop("PGH_CHEB_CSR", None)
                if blocks is not None:
                    os.environ["PGH_BLOCKS64"] = blocks
                if route == "csr":
                    os.environ["PGH_CHEB_CSR"] = "1"
                ranker = pg.HeatKernel(5, coefficient_type="chebyshev", error_type=pg.L1, tol=1e-9, max_iters=40,
                                       preprocessor=pg.preprocessor(normalization=normalization, assume_immutability=False))
                got = np.asarray(ranker.rank(pg.AdjacencyWrapper(graph, directed=True), p.copy()).np, dtype=np.float64)
                assert ranker.convergence.iteration == want_iters, (label, route)
                assert np.max(np.abs(got - want)) / np.max(np.abs(want)) <= 1e-6, (label, route)
                results[route] = got
            assert np.max(np.abs(results["blocked"] - results["csr"])) / np.max(np.abs(want)) <= 1e-7, label
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_f64_cold_image(pg):
    """The cold tail of the f64 image in its own propagation-blocking image (pgh_bsf64.hip, round 6: k_pb64_gather / k_pb64_finish,
    hot-only 2-byte stream) -- forced on a graph of 1 M nodes (large graphs get it by default): the "chebyshev" recurrence
    (abstract_filters.py:216-224), PageRank and AbsorbingWalks with f64 iterates (tol = 1e-9: adhoc.py:34-36, 166-169) against the
    oracle with equal iteration counts, under every layout switch -- the image alone, hub bins + rows left in the stream, 4-byte stream
    words, the cold gathers left in the stream (round 5's route) -- value-free and valued, and the routes against each other."""
    import os
    from pygrank_amd import _lib as L
    if not L.runtime_name().startswith("hip:"):
        return
    from oracle import ref_loops as orc
    rng = np.random.default_rng(5)
    A = rmat_np.rmat_csr(20, 16, seed=2)                       # 1 M nodes, half of them live: 8 blocks of ~65 K referenced sources, 20 224 of each in the hot cache
    W = sp.csr_array(A)
    W.data = rng.uniform(0.5, 2.0, W.nnz)
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 60, seed=3)] = rng.uniform(0.5, 1.5, 60)
    keys = ("PGH_PB", "PGH_PB_FORCE", "PGH_PB_HEAVY", "PGH_PB_HUBMAX", "PGH_PB64", "PGH_STREAM16")
    saved = {k: os.environ.get(k) for k in keys}
    layouts = (("cold image", dict(PGH_PB="1", PGH_PB_FORCE="1"), "f64 propagation-blocking", "(2 B/entry)"),
               ("hub bins, heavy rows in the stream", dict(PGH_PB="1", PGH_PB_FORCE="1", PGH_PB_HEAVY="512", PGH_PB_HUBMAX="300"),
                "heavy rows stay in the stream", None),
               ("4-byte stream words", dict(PGH_PB="1", PGH_PB_FORCE="1", PGH_STREAM16="0"), "f64 propagation-blocking", "(4 B/entry)"),
               ("cold gathers in the stream", dict(PGH_PB64="0"), None, None))
    try:
        for label, graph in (("value-free", A), ("valued", W)):
            M = orc.normalize(graph, "col", True)
            want = {"cheb": orc.heat_kernel(M, p, t=5, coefficient_type="chebyshev", tol=1e-9, max_iters=40, error_type="l1"),
                    "ppr": orc.pagerank(M, p, alpha=0.85, error_type="l1", tol=1e-9, max_iters=200),
                    "absorb": orc.absorbing_walks(M, p, alpha=0.85, error_type="l1", tol=1e-9, max_iters=200)}
            got = {}
            for name, env, must_have, stream in layouts:
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(env)
                adj = pg.preprocessor(normalization="col", assume_immutability=True)(pg.AdjacencyWrapper(graph, directed=True))
                rankers = {"cheb": pg.HeatKernel(5, coefficient_type="chebyshev", error_type=pg.L1, tol=1e-9, max_iters=40),
                           "ppr": pg.PageRank(0.85, error_type=pg.L1, tol=1e-9, max_iters=200),
                           "absorb": pg.AbsorbingWalks(0.85, error_type=pg.L1, tol=1e-9, max_iters=200)}
                for which, ranker in rankers.items():
                    res = np.asarray(ranker.rank(adj, p.copy()).np, dtype=np.float64)
                    assert ranker.convergence.iteration == want[which][1], (label, name, which, ranker.convergence.iteration, want[which][1])
                    assert np.max(np.abs(res - want[which][0])) / np.max(np.abs(want[which][0])) <= 1e-6, (label, name, which)
                    got[(name, which)] = res
                fmt = adj.array.format()
                assert "f64 image" in fmt, fmt
                if label == "value-free":
                    assert (must_have is None) == ("f64 propagation-blocking" not in fmt), (name, fmt)
                    assert must_have is None or must_have in fmt, (name, fmt)
                    assert stream is None or stream in fmt.split("f64 image")[1], (name, fmt)
            for which in ("cheb", "ppr", "absorb"):                # the four layouts agree far below the f32 result's resolution
                ref = got[("cold gathers in the stream", which)]
                for name, _, _, _ in layouts[:3]:
                    assert np.max(np.abs(got[(name, which)] - ref)) <= 2e-7 * np.max(np.abs(ref)), (label, name, which)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_loop_is_deterministic(pg):
    A = rmat_np.rmat_csr(14, 8, seed=2)
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 30, seed=1)] = 1.0
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(assume_immutability=True)
    runs = [np.asarray(pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=500).rank(graph, p.copy()).np,
                       dtype=F32) for _ in range(3)]
    assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2])   # atomic-free, fixed order


def check_small_graph_tail_launch(pg):
    """Graphs of up to 12 K rows close a recursive step with ONE one-workgroup launch (k_small_tail: fix-ups, epilogue,
    residual, ConvergenceManager) instead of three; PGH_SMALL_TAIL=0 keeps the general sequence.  Both against the oracle
    (1e-6 of the largest rank, equal iteration counts) and against each other, for every stopping rule, with and without the
    quotient, for AbsorbingWalks and the taylor form of HeatKernel, on row counts around the 1024-thread rounds of the kernel."""
    import os
    saved = os.environ.get("PGH_SMALL_TAIL")
    try:
        for scale, ef, seed in ((6, 4, 0), (10, 8, 1), (12, 8, 2), (13, 6, 3)):
            A = rmat_np.rmat_csr(scale, ef, seed=seed)
            if scale == 13:
                A = sp.csr_array(A[:8191, :8191])                 # not a multiple of anything
            n = A.shape[0]
            p = np.zeros(n)
            p[rmat_np.seed_nodes(A, 5, seed=seed)] = np.arange(1, 6)
            M = sp.csr_array(orc.normalize(A, "col", True))
            graph = pg.AdjacencyWrapper(A, directed=True)
            cases = [("pagerank", dict(error_type="l1", tol=1e-6, max_iters=500)), ("pagerank", dict(error_type="mabs", tol=1e-8, max_iters=500)),
                     ("pagerank", dict(error_type="linf", tol=1e-7, max_iters=500)), ("pagerank", dict(error_type="iters", max_iters=17)),
                     ("pagerank", dict(error_type="l1", tol=1e-6, max_iters=500, use_quotient=False)),
                     ("absorbing", dict(error_type="l1", tol=1e-6, max_iters=500)),
                     ("heat", dict(error_type="l1", tol=1e-6, max_iters=100)), ("heat", dict(error_type="linf", tol=1e-5, max_iters=100)),
                     ("heat", dict(error_type="iters", max_iters=12))]       # (stopping rules well above the f32 resolution of the result)
            for name, kw in cases:
                if name == "pagerank":
                    want, want_iters = orc.pagerank(M, p, alpha=0.85, eps=EPS32, **kw)
                elif name == "heat":
                    want, want_iters = orc.heat_kernel(M, p, t=3, eps=EPS32, **kw)
                else:
                    want, want_iters = orc.absorbing_walks(M, p, alpha=0.85, eps=EPS32, **kw)
                outcomes = []
                measure = {"l1": pg.L1, "mabs": pg.Mabs, "linf": pg.MaxDifference, "iters": "iters"}[kw["error_type"]]
                for switch in ("1", "0"):
                    os.environ["PGH_SMALL_TAIL"] = switch
                    pre = pg.preprocessor(assume_immutability=True, normalization="col")
                    opts = dict(kw, error_type=measure, preprocessor=pre, dtype="float32")      # (the f32 kernels are what is checked here)
                    algo = (pg.PageRank(0.85, **opts) if name == "pagerank" else pg.HeatKernel(3, **opts) if name == "heat"
                            else pg.AbsorbingWalks(0.85, **opts))
                    got = _np(algo.rank(graph, p.copy()).np)
                    assert algo.convergence.iteration == want_iters, (scale, name, kw, switch, algo.convergence.iteration, want_iters)
                    assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want)), (scale, name, kw, switch)
                    outcomes.append(got)
                assert np.max(np.abs(outcomes[0] - outcomes[1])) <= 4 * EPS32 * np.max(np.abs(want)), (scale, name, kw)
    finally:
        if saved is None:
            os.environ.pop("PGH_SMALL_TAIL", None)
        else:
            os.environ["PGH_SMALL_TAIL"] = saved


def check_error_reporting(pg):
    import pytest
    from pygrank_amd import _lib as L
    a, b = _vec(pg, [1, 2, 3]), _vec(pg, [1, 2])
    with pytest.raises(L.EngineError):
        a + b
    g = pg.scipy_sparse_to_backend(sp.csr_array(np.eye(3)))
    with pytest.raises(L.EngineError):
        pg.conv(b, g)
    with pytest.raises(L.EngineError):
        a[7]


def check_rmat_generator_matches_numpy(pg):
    """Device RMAT generation + on-GPU normalisation == oracle/rmat_np.py + the reference's host normalisation
    (preprocessing.py:109-113,131-138): structure bit-exact, values equal after the single rounding to f32."""
    from oracle import ref_loops as orc
    from pygrank_amd.synthetic import rmat_device_graph, rmat_graph
    for scale, ef, seed in ((6, 4, 1), (11, 8, 0), (14, 16, 3)):
        A = rmat_np.rmat_csr(scale, ef, seed=seed)
        for normalization, sym in (("none", False), ("col", False), ("symmetric", False), ("symmetric", True), ("col", True)):
            B = sp.csr_array(A + A.T) if sym else A
            M = sp.csr_array(orc.normalize(B, normalization, True))
            g = rmat_device_graph(scale, ef, seed=seed, normalization=normalization, symmetrize=sym)
            assert g.shape == M.shape and g.nnz == M.nnz, (scale, normalization, sym)
            MT = g.download_transposed()
            want = sp.csr_array(M.T)
            want.sort_indices()
            assert np.array_equal(MT.indptr, want.indptr) and np.array_equal(MT.indices, want.indices)
            assert np.allclose(MT.data, want.data, rtol=1.5 * EPS32, atol=0), (scale, normalization, sym)
            if normalization == "none":
                assert np.array_equal(MT.data, want.data.astype(F32))          # integer multiplicities: exact
            assert np.allclose(_np(pg.degrees(g)), orc.row_sums(M), rtol=2 * EPS32, atol=1e-30)
    # row slices (1-D partition of M^T) stack back to the full matrix
    A = rmat_np.rmat_csr(11, 8, seed=0)
    M = sp.csr_array(orc.normalize(A, "col", True))
    full = sp.csr_array(M.T)
    bounds = [0, 300, 301, 1500, 2048]
    parts = [rmat_device_graph(11, 8, seed=0, normalization="col", row_begin=b0, row_end=b1).download_transposed()
             for b0, b1 in zip(bounds[:-1], bounds[1:])]
    stacked = sp.vstack(parts).tocsr()
    assert stacked.shape == full.shape and abs(stacked - full).max() <= 1.5 * EPS32
    # the preprocessed graph object plugs into the filters like a preprocessor outcome
    adj = rmat_graph(11, 8, seed=0)
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 20, seed=1)] = 1.0
    ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=500)
    got = np.asarray(ranker.rank(adj, p.copy()).np)
    want, it = orc.pagerank(M, p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=500)
    assert ranker.convergence.iteration == it
    assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))


def check_multi_seed_spmm_and_batched_pagerank(pg):
    """One pass over the adjacency for a slab of seeds == the per-column single-vector results (signals.py:225-226)."""
    from oracle import ref_loops as orc
    rng = np.random.default_rng(9)
    for name, M in matrices():
        if M.shape[0] != M.shape[1] or M.shape[0] < 2:
            continue
        n = M.shape[0]
        g = pg.scipy_sparse_to_backend(M)
        M32 = sp.csr_array(M.astype(F32).astype(np.float64))
        for b in (1, 3, 16, 17, 24, 32, 33, 61, 64):     # the lane shapes of the multi-seed kernels: 4 / 8 / 16 lanes per row
            X = rng.random((n, b)).astype(F32).astype(np.float64)
            Y = np.asarray(pg.conv(pg.to_primitive(X), g))
            ref = (X.T @ M32).T
            scale = (np.abs(X).T @ np.abs(M32)).T
            assert Y.shape == (n, b), name
            assert np.all(np.abs(Y - ref) <= 4 * EPS32 * scale + 1e-30), (name, b)
    # batched PageRank: every column stops at its own iteration and equals the single-vector run / the oracle
    A = rmat_np.rmat_csr(12, 8, seed=5)
    n = A.shape[0]
    Mn = sp.csr_array(orc.normalize(A, "col", True))
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(assume_immutability=True)
    feats = np.zeros((n, 5))
    for j in range(5):
        feats[rmat_np.seed_nodes(A, [1, 2000, 40, 5, 300][j], seed=j + 1), j] = 1.0 + j
    feats[:, 3] = 0.0                                               # a zero personalization column stays zero
    ranker = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=500)
    out = np.asarray(ranker.propagate(graph, pg.to_primitive(feats)))
    assert out.shape == (n, 5) and hasattr(ranker, "last_batches")
    iters = [c["iterations"] for c in ranker.last_batches[0]]
    borderline = set()                       # columns whose residual sits within f32 rounding of the tolerance at the stop
    for j in range(5):
        if j == 3:
            assert np.all(out[:, j] == 0)
            continue
        want, it = orc.pagerank(Mn, feats[:, j], alpha=0.85, error_type="l1", tol=1e-6, max_iters=500)
        if iters[j] != it:
            # An iteration apart is accepted ONLY when the oracle's own residual at the engine's stopping check lies within
            # 2 % of the tolerance: the f32 rounding of a column that sits on one heavy seed node moves its L1 residual by
            # ~1e-8 (column 0: 1.0030e-6 at iteration 13 against tol = 1e-6).  The result is then held to the oracle stopped
            # at the engine's count.
            assert abs(iters[j] - it) == 1, (j, iters[j], it)
            stop = iters[j]
            at_stop = orc.pagerank(Mn, feats[:, j], alpha=0.85, error_type="iters", max_iters=stop)[0]
            before = orc.pagerank(Mn, feats[:, j], alpha=0.85, error_type="iters", max_iters=stop - 1)[0]
            residual = np.abs(at_stop - before).sum() / np.abs(feats[:, j]).sum()
            assert abs(residual - 1e-6) <= 2e-8, (j, iters[j], it, residual)
            want = at_stop
            borderline.add(j)
        assert np.max(np.abs(out[:, j] - want)) <= 1e-6 * np.max(np.abs(want)), j
    assert len(set(i for k, i in enumerate(iters) if k != 3)) > 1, iters   # the columns really stop at different iterations
    # the same seeds four times over (20 columns: 8 lanes per row instead of 4): same stopping iterations, same columns
    wide = np.asarray(ranker.propagate(graph, pg.to_primitive(np.tile(feats, (1, 4)))))
    wide_iters = [c["iterations"] for c in ranker.last_batches[0]]
    assert all(abs(wide_iters[j] - iters[j % 5]) <= (1 if j % 5 in borderline else 0) for j in range(20)), (wide_iters, iters)
    for j in range(20):
        assert np.max(np.abs(wide[:, j] - out[:, j % 5])) <= 1e-6 * max(np.max(np.abs(out[:, j % 5])), 1e-30), j


def check_factored_upload_matches_valued_upload(pg):
    """pgh_graph_from_factored_csr(W, left, right) stores the same matrix as pgh_graph_from_csr(diag(left) W diag(right)):
    bit-identical f32 values / degrees, and the same propagation results whether the engine picked the value-free
    layout (integer multi-edge weights) or the valued one (real weights)."""
    from pygrank_amd.device import DeviceGraph
    from pygrank_amd.preprocessing import normalize_adjacency
    rng = np.random.default_rng(11)
    cases = []
    A = rmat_np.rmat_csr(13, 8, seed=3)                             # integer multiplicities, dangling + empty rows
    cases.append(("rmat13_int", A))
    cases.append(("rmat13_sym", sp.csr_array(A + A.T)))
    B = _random_matrix(rng, 2000, 2000, 0.004, hubs=3, empty_frac=0.2)
    cases.append(("real_weights", B))
    C = sp.csr_array(_random_matrix(rng, 300, 500, 0.05))
    C.data[:] = 1.0
    cases.append(("rect_unit", C))
    big = sp.csr_array(A.copy())
    big.data[::7] = 40000.0                                         # multiplicity too large to expand: valued layout
    cases.append(("heavy_int", big))
    for name, W in cases:
        for norm in ("col", "symmetric", "both", "none"):
            if norm != "none" and W.shape[0] != W.shape[1]:
                continue                                            # the reference's normalisations are square-only
            N = normalize_adjacency(W, norm)
            if norm == "none":
                gf = DeviceGraph.from_factored(W)
            else:
                assert hasattr(N, "_pgh_factors"), (name, norm)
                gf = pg.scipy_sparse_to_backend(N)                  # routed through the factors
                if name.startswith("rmat13"):
                    assert "value-free" in gf.format() or "host" in gf.format(), gf.format()
            plain = sp.csr_array((N.data.copy(), N.indices.copy(), N.indptr.copy()), shape=N.shape)
            gv = pg.scipy_sparse_to_backend(plain)                  # valued upload of the host-evaluated product
            a, b = gf.download_transposed(), gv.download_transposed()
            assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), (name, norm)
            assert np.array_equal(a.data, b.data), (name, norm, float(np.max(np.abs(a.data - b.data))))
            assert np.allclose(_np(pg.degrees(gf)), _np(pg.degrees(gv)), rtol=EPS32, atol=0), (name, norm)
            x = rng.random(W.shape[0]).astype(F32).astype(np.float64)
            yf, yv = _np(pg.conv(_vec(pg, x), gf)), _np(pg.conv(_vec(pg, x), gv))
            scale = np.abs(x) @ np.abs(sp.csr_array(N))
            assert np.all(np.abs(yf - yv) <= 6 * EPS32 * scale + 1e-30), (name, norm)
    # rectangular with explicit scales on both sides
    left, right = rng.random(C.shape[0]) + 0.5, rng.random(C.shape[1]) + 0.5
    gf = DeviceGraph.from_factored(C, left, right)
    N = sp.csr_array(sp.diags(left) @ C @ sp.diags(right))
    gv = pg.scipy_sparse_to_backend(N)
    assert np.array_equal(gf.download_transposed().data, gv.download_transposed().data)
    x = rng.random(C.shape[0]).astype(F32).astype(np.float64)
    yf, yv = _np(pg.conv(_vec(pg, x), gf)), _np(pg.conv(_vec(pg, x), gv))
    assert yf.shape == (C.shape[1],) and np.all(np.abs(yf - yv) <= 6 * EPS32 * (np.abs(x) @ np.abs(N)) + 1e-30)
    # end to end through the preprocessor: PageRank on the multigraph equals the oracle
    from oracle import ref_loops as orc
    graph = pg.AdjacencyWrapper(A, directed=True)
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 50, seed=4)] = 1.0
    ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=500)
    got = np.asarray(ranker.rank(graph, p).np)
    want, it = orc.pagerank(sp.csr_array(orc.normalize(A, "col", True)), p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=500)
    assert ranker.last_loop["iterations"] == it
    assert np.max(np.abs(got - want)) <= 1e-6 * np.max(np.abs(want))


def check_device_preprocessor_matches_host(pg):
    """pgh_graph_from_adjacency (normalisation evaluated in HBM) == the reference's host normalisation
    followed by an upload: identical structure; identical f32 values for integer weights (the degree sums are exact),
    within 1 ulp of f32 for real weights (scipy sums a row in a different order)."""
    from pygrank_amd.device import DeviceGraph
    from pygrank_amd.preprocessing import normalize_adjacency, to_sparse_matrix
    rng = np.random.default_rng(17)
    A = rmat_np.rmat_csr(13, 8, seed=3)
    unit = sp.csr_array(A.copy())
    unit.data[:] = 1.0
    real = _random_matrix(rng, 1500, 1500, 0.006, hubs=3, empty_frac=0.2)
    rect = sp.csr_array(_random_matrix(rng, 200, 700, 0.03))
    for name, W, exact in (("rmat13_int", A, True), ("rmat13_sym", sp.csr_array(A + A.T), True), ("unit", unit, True),
                           ("real", real, False), ("rect", rect, False), ("empty", sp.csr_array((6, 6)), True)):
        for norm in ("col", "symmetric", "both", "none"):
            if W.shape[0] != W.shape[1] and norm != "none":
                continue                                            # the reference's normalisations are square-only
            gd = DeviceGraph.from_adjacency(W, norm)
            N = normalize_adjacency(W, norm)
            plain = sp.csr_array((N.data.copy(), N.indices.copy(), N.indptr.copy()), shape=N.shape)
            gh = pg.scipy_sparse_to_backend(plain)
            a, b = gd.download_transposed(), gh.download_transposed()
            assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices), (name, norm)
            if exact:
                assert np.array_equal(a.data, b.data), (name, norm)
            else:
                assert np.allclose(a.data, b.data, rtol=1.01 * EPS32, atol=0), (name, norm)
            assert np.allclose(_np(pg.degrees(gd)), _np(pg.degrees(gh)), rtol=2 * EPS32, atol=1e-30), (name, norm)
            if W.nnz and name.startswith(("rmat13", "unit")) and "host" not in gd.format():
                assert "value-free" in gd.format(), (name, norm, gd.format())
    # the preprocessor takes the device route for every named normalisation -- round 6: the renormalisation trick (W + r I,
    # preprocessing.py:107-108) and the laplacian (I - N, :114-122) included, whose diagonal entries the device APPENDS to the rows -- and
    # the host route for callables / reductions / transforms.  Same matrix either way: the device's image with its duplicate positions
    # summed against the host route's (scipy merges them), <= 1 ulp of f32 where two f32 entries were added; same degrees and products.
    graph = pg.AdjacencyWrapper(A, directed=True)
    x = np.linspace(0.1, 1.0, A.shape[0])
    for kw in (dict(normalization="col"), dict(normalization="symmetric"), dict(normalization="col", renormalize=True),
               dict(normalization="laplacian"), dict(normalization="symmetric", renormalize=True), dict(normalization="both", renormalize=0.5),
               dict(normalization="laplacian", renormalize=True), dict(normalization="none", renormalize=2)):
        dev_adj = to_sparse_matrix(graph, **kw)
        host_adj = to_sparse_matrix(graph, transform_adjacency=lambda m: m, **kw)
        dev, host = sp.csr_array(dev_adj.array.download_transposed().astype(np.float64)), sp.csr_array(host_adj.array.download_transposed().astype(np.float64))
        added = bool(kw.get("renormalize")) or kw["normalization"] == "laplacian"
        assert dev.nnz == host.nnz + (0 if not added else dev.nnz - host.nnz) and (added or dev.nnz == host.nnz), kw
        dev.sum_duplicates()
        host.sum_duplicates()
        dev.sort_indices()
        host.sort_indices()
        dev.eliminate_zeros()
        host.eliminate_zeros()
        assert np.array_equal(dev.indptr, host.indptr) and np.array_equal(dev.indices, host.indices), kw
        assert np.allclose(dev.data, host.data, rtol=(2.5 * EPS32 if added else 0.0), atol=0), (kw, float(np.max(np.abs(dev.data - host.data))))
        assert np.allclose(_np(pg.degrees(dev_adj)), _np(pg.degrees(host_adj)), rtol=4 * EPS32, atol=4 * EPS32), kw
        assert np.allclose(_np(pg.conv(pg.to_array(x), dev_adj)), _np(pg.conv(pg.to_array(x), host_adj)), rtol=1e-5, atol=1e-6), kw
    # ... and against the REFERENCE's own preprocessor (tests/golden/golden_norm.npz: its outputs for 5 normalisations x renormalize on
    # two graphs, generated from /root/reference by tests/golden/make_golden.py)
    import os
    import cases
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_norm.npz"))
    for gkey in ("rmat10_dir", "weighted300"):
        Ag, directed, _ = cases.GRAPHS[gkey]()
        xg = np.linspace(0.1, 1.0, Ag.shape[0])
        for normalization in cases.NORMALIZATIONS:
            for renorm in (False, True):
                key = f"{gkey}|{normalization}|{int(renorm)}"
                adj = pg.preprocessor(normalization=normalization, renormalize=renorm)(pg.AdjacencyWrapper(Ag, directed=directed))
                want = sp.csr_array((gold[key + "|data"], gold[key + "|indices"], gold[key + "|indptr"]), shape=Ag.shape)
                got = sp.csr_array(adj.array.download_transposed().astype(np.float64).T)
                got.sum_duplicates()
                diff = abs(got - want)
                assert (diff.max() if diff.nnz else 0.0) <= 2.5 * EPS32 * max(abs(want).max(), 1e-30), (key, diff.max())
                assert np.allclose(_np(pg.degrees(adj)), gold[key + "|degrees"], rtol=4 * EPS32, atol=4 * EPS32), key
                assert np.allclose(_np(pg.conv(pg.to_array(xg), adj)), gold[key + "|conv"], rtol=1e-5, atol=1e-6), key


def check_propagation_blocking_image(pg):
    """The propagation-blocking image of the cold tail (csrc/pgh_pb.hip; forced here, large graphs get it by default):
    same products and same PageRank as with the cold gathers left in the stream, for value-free and valued graphs,
    deterministic, and exact to f32 rounding over 30 decades of input magnitudes (its row sums are 64-bit fixed point).
    (No-op on the test double.)"""
    import os
    from pygrank_amd import _lib as L
    if not L.runtime_name().startswith("hip:"):
        return
    from pygrank_amd.device import DeviceGraph
    rng = np.random.default_rng(23)
    A = rmat_np.rmat_csr(17, 16, seed=1)                        # 131 K nodes: four times the LDS hot cache
    Wreal = sp.csr_array(A.copy())
    Wreal.data = Wreal.data * (0.5 + rng.random(Wreal.nnz))
    saved = {k: os.environ.get(k) for k in ("PGH_PB", "PGH_PB_FORCE", "PGH_PB_HEAVY", "PGH_PB_HUBMAX")}
    os.environ["PGH_PB_HEAVY"], os.environ["PGH_PB_HUBMAX"] = "256", "2000"       # hub bins and rows left in the stream, both
    try:
        for name, W in (("int", A), ("real", Wreal)):
            os.environ["PGH_PB"] = "0"
            os.environ.pop("PGH_PB_FORCE", None)
            g0 = DeviceGraph.from_adjacency(W, "col")
            os.environ["PGH_PB"], os.environ["PGH_PB_FORCE"] = "1", "1"
            g1 = DeviceGraph.from_adjacency(W, "col")
            assert "propagation-blocking" in g1.format() and "propagation-blocking" not in g0.format(), (g0.format(), g1.format())
            x = rng.random(W.shape[0]).astype(F32).astype(np.float64)
            y0, y1 = _np(pg.conv(_vec(pg, x), g0)), _np(pg.conv(_vec(pg, x), g1))
            N = sp.csr_array(g0.download_transposed().T)
            scale = np.abs(x) @ np.abs(N)
            assert np.all(np.abs(y0 - y1) <= 8 * EPS32 * scale + 1e-30), name
            assert np.array_equal(_np(pg.conv(_vec(pg, x), g1)), y1), name          # deterministic
            # signed values over 30 decades: every row stays within f32 rounding of the exact sum, plus the fixed-point
            # quantum of the launch (<= 2^-48 of the largest |value| per entry)
            xw = (rng.choice([-1.0, 1.0], W.shape[0]) * 10.0 ** rng.uniform(-30, 0, W.shape[0])).astype(F32).astype(np.float64)
            yw = _np(pg.conv(_vec(pg, xw), g1))
            exact = xw @ N
            bound = 8 * EPS32 * (np.abs(xw) @ np.abs(N)) + 1e-11 * np.max(np.abs(xw)) * np.max(np.abs(N.data))
            assert np.all(np.abs(yw - exact) <= bound), (name, float(np.max(np.abs(yw - exact) / bound)))
            p = np.zeros(W.shape[0])
            p[rmat_np.seed_nodes(A, 50, seed=4)] = 1.0
            from pygrank_amd.preprocessing import Adjacency
            from pygrank_amd.signals import _IdentityMap
            runs = []
            for g in (g0, g1):
                ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=500)
                adj = Adjacency(g)
                adj._pygrank_preprocessed = {"hip": adj}
                adj._pygrank_node2id = _IdentityMap(g.shape[0])
                adj.is_directed = lambda: True
                r = ranker.rank(adj, p.copy())
                runs.append((np.asarray(r.np), ranker.last_loop["iterations"]))
            assert runs[0][1] == runs[1][1], name
            assert np.max(np.abs(runs[0][0] - runs[1][0])) <= 1e-6 * np.max(np.abs(runs[0][0])), name
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def check_slab_ops_and_wide_propagate(pg):
    """Whole-slab prologue kernels (pgh_mat_col_abssum / div_cols / get_cols / set_cols) against numpy, and propagate
    over more than 64 feature columns (two batches) against per-column rank()."""
    from pygrank_amd.device import DeviceMatrix
    rng = np.random.default_rng(21)
    for n, b in ((1, 1), (1000, 3), (4097, 64), (513, 70), (70001, 17), (300007, 3)):   # the last two: several grid periods, b not a divisor
        X = (rng.random((n, b)) - 0.3).astype(F32).astype(np.float64)
        X[:, b // 2] = 0.0
        D = DeviceMatrix.from_host(X)
        sums = D.col_abssum()
        assert np.allclose(sums, np.abs(X).sum(axis=0), rtol=1e-12, atol=0), (n, b)
        assert np.array_equal(D.col_abssum(), sums)                   # deterministic
        Q = np.asarray(D.div_cols(sums))
        want = X.copy()
        nz = sums != 0
        want[:, nz] = (X[:, nz].astype(F32) / sums[nz].astype(F32)).astype(np.float64)
        assert np.allclose(Q, want, rtol=2 * EPS32, atol=0), (n, b)
        if b > 2:
            part = D.get_cols(1, b - 2)
            assert np.array_equal(np.asarray(part), X[:, 1:b - 1])
            Z = DeviceMatrix.from_host(np.zeros((n, b)))
            Z.set_cols(1, part)
            back = np.asarray(Z)
            assert np.array_equal(back[:, 1:b - 1], X[:, 1:b - 1]) and np.all(back[:, 0] == 0) and np.all(back[:, -1] == 0)
    A = rmat_np.rmat_csr(10, 8, seed=7)
    n = A.shape[0]
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(assume_immutability=True)
    feats = np.zeros((n, 70))
    for j in range(70):
        feats[rmat_np.seed_nodes(A, 3 + j, seed=j + 1), j] = 1.0 + 0.1 * j
    feats[:, 66] = 0.0
    ranker = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=500)
    out = np.asarray(ranker.propagate(graph, pg.to_primitive(feats)))
    assert out.shape == (n, 70) and len(ranker.last_batches) == 2 and len(ranker.last_batches[1]) == 6
    for j in (0, 17, 63, 64, 66, 69):
        single = np.asarray(ranker.rank(graph, feats[:, j]).np)
        assert np.max(np.abs(out[:, j] - single)) <= 2e-6 * max(np.max(np.abs(single)), 1e-30), j


def check_trimmed_gather_layout(pg):
    """pgh_graph_gather_layout / pgh_graph_set_gather_bases: a partitioned step gives bit-identical results whether the
    gather vector is stored in full (block b at b * blk) or trimmed to the referenced prefix of every block."""
    import ctypes as C
    from pygrank_amd import _lib as L
    from pygrank_amd.device import DeviceVector
    from pygrank_amd.distributed import rmat_partitioned
    lib = L.lib()
    rng = np.random.default_rng(13)
    for scale, ef in ((12, 4), (15, 8)):
        part = rmat_partitioned(scale, ef, 0, 1)
        g, n = part.graph, part.n
        nb, blk = C.c_int32(), C.c_int64()
        live = np.zeros(8, dtype=np.int32)
        L.check(lib.pgh_graph_gather_layout(g._h, C.byref(nb), C.byref(blk), live.ctypes.data_as(C.c_void_p)))
        nb, blk = nb.value, blk.value
        assert nb * blk == n and np.all(live[:nb] >= 1) and np.all(live[:nb] <= blk) and np.all(live[nb:] == 0)
        assert live[:nb].max() < blk          # RMAT: a large share of the sources is never referenced
        x = rng.random(n).astype(F32)
        p = rng.random(n).astype(F32)
        xg_local, dx = DeviceVector.empty(n), DeviceVector.from_host(x.astype(np.float64))
        dp = DeviceVector.from_host(p.astype(np.float64))
        L.check(lib.pgh_dist_prescale(g._h, dx._h, xg_local._h))
        xg = np.asarray(xg_local).astype(F32)

        def step(layout_bases, gather):
            bases = np.zeros(8, dtype=np.int64)
            bases[:nb] = layout_bases
            L.check(lib.pgh_graph_set_gather_bases(g._h, bases.ctypes.data_as(C.c_void_p)))
            y, xo, s = DeviceVector.empty(n), DeviceVector.empty(n), C.c_double()
            dg = DeviceVector.from_host(gather.astype(np.float64))
            L.check(lib.pgh_ppr_step_dist(g._h, dg._h, 0.7, dp._h, 0.85, y._h, xo._h, C.byref(s)))
            return np.asarray(y).astype(F32), np.asarray(xo).astype(F32), s.value

        full = step(np.arange(nb) * blk, xg)
        top = int((live[:nb].max() + 63) // 64 * 64)
        compact = np.zeros(nb * top + 32768, dtype=F32)
        for b in range(nb):
            compact[b * top:b * top + min(top, blk)] = xg[b * blk:b * blk + min(top, blk)]
        trimmed = step(np.arange(nb) * top, compact)
        assert np.array_equal(full[0], trimmed[0]) and np.array_equal(full[1], trimmed[1]) and full[2] == trimmed[2]
        # a gather vector shorter than the layout is refused
        short = DeviceVector.from_host(np.zeros(top, dtype=np.float64))
        y, xo = DeviceVector.empty(n), DeviceVector.empty(n)
        rc = lib.pgh_ppr_step_dist(g._h, short._h, 1.0, dp._h, 0.85, y._h, xo._h, None)
        if L.runtime_name().startswith("hip:"):
            assert rc != 0 and b"gather vector" in lib.pgh_last_error()


ALL = [v for k, v in sorted(globals().items()) if k.startswith("check_") and callable(v)]
