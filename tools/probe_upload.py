"""Uploaded-graph path: scipy adjacency -> preprocessor -> factored upload -> PPR.  PGH_VALUES=1 forces the valued layout."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygrank_amd as pg
from pygrank_amd import _lib as L
from oracle import rmat_np

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=21)
ap.add_argument("--ef", type=int, default=16)
ap.add_argument("--runs", type=int, default=10)
args = ap.parse_args()
pg.load_backend("hip")
t0 = time.time()
A = rmat_np.rmat_csr(args.scale, args.ef, seed=0)
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
    p[rmat_np.seed_nodes(A, 100, seed=1)] = 1.0
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
