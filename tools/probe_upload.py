"""Uploaded-graph path: scipy adjacency -> preprocessor -> factored upload -> PPR.  PGH_VALUES=1 forces the valued layout."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygrank_amd as pg
from pygrank_amd import _lib as L



def host_rmat(scale, ef, a=0.57, b=0.19, c=0.19, seed=0):
    """A host-side RMAT multigraph (numpy RNG; duplicates summed into integer weights) as scipy CSR: the kind of input
    a pygrank user hands to the preprocessor."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n, m = 1 << scale, (1 << scale) * ef
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(m)
        src = (src << 1) | (r >= a + b)
        dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    return sp.csr_array(sp.coo_array((np.ones(m), (src, dst)), shape=(n, n)))


ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--ef", type=int, default=16)
ap.add_argument("--runs", type=int, default=10)
args = ap.parse_args()
pg.load_backend("hip")
t0 = time.time()
A = host_rmat(args.scale, args.ef)
t_gen = time.time() - t0
for values in ("0", "1"):
    os.environ["PGH_VALUES"] = values
    graph = pg.AdjacencyWrapper(A, directed=True)
    pre = pg.preprocessor(assume_immutability=True)
    t0 = time.time()
    adj = pre(graph)
    L.check(L.lib().pgh_sync())
    t_up = time.time() - t0
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    p = np.zeros(n)
    p[np.random.default_rng(1).choice(np.flatnonzero(np.diff(A.indptr) > 0), 100, replace=False)] = 1.0
    sig = pg.to_signal(adj, p)
    ranker = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=1000)
    ranker.rank(adj, sig)
    L.check(L.lib().pgh_sync())
    t0 = time.perf_counter()
    spmv = 0
    for _ in range(args.runs):
        ranker.rank(adj, sig)
        spmv += ranker.last_loop["spmv"]
    L.check(L.lib().pgh_sync())
    dt = time.perf_counter() - t0
    print(f"scale={args.scale} n={n} nnz={nnz} host_gen={t_gen:.1f}s preprocess+upload={t_up:.2f}s "
          f"iters={ranker.last_loop['iterations']} GTEPS={nnz * spmv / dt / 1e9:.1f} PGH_VALUES={os.environ.get('PGH_VALUES', '0')} format={g.format()}")
