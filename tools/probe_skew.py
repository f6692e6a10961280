"""PPR throughput as a function of the degree skew of the input (RMAT a/b/c): the LDS hot cache only helps when a few
sources collect most references.  Usage: python tools/probe_skew.py --scale 23"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pygrank_amd as pg  # noqa: E402
from pygrank_amd import _lib as L  # noqa: E402
from pygrank_amd.synthetic import rmat_graph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=23)
args = ap.parse_args()
pg.load_backend("hip")
for a, b, c in ((0.57, 0.19, 0.19), (0.45, 0.22, 0.22), (0.35, 0.25, 0.25), (0.25, 0.25, 0.25)):
    adj = rmat_graph(args.scale, 16, a=a, b=b, c=c, seed=0, normalization="col")
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    deg = np.asarray(pg.degrees(g))
    p = np.zeros(n)
    p[np.random.default_rng(1).choice(np.flatnonzero(deg > 0), 100, replace=False)] = 1.0
    sig = pg.to_signal(adj, p)
    ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000)
    ranker.rank(adj, sig)
    L.check(L.lib().pgh_sync())
    t0 = time.perf_counter()
    spmv, loop_ms = 0, 0.0
    for _ in range(5):
        ranker.rank(adj, sig)
        spmv += ranker.last_loop["spmv"]
        loop_ms += ranker.last_loop["loop_ms"]
    L.check(L.lib().pgh_sync())
    dt = time.perf_counter() - t0
    print(f"rmat a={a} b={b} c={c}: nnz={nnz} live-out-degree nodes={int((deg > 0).sum())} iterations={ranker.last_loop['iterations']} "
          f"GTEPS={nnz * spmv / dt / 1e9:6.1f} device step={loop_ms / spmv * 1e3:6.1f} us ({g.format().split(',')[0]})", flush=True)
    del adj, g, sig
