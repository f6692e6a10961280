"""PageRank with the reference's DEFAULT stopping rule (Mabs, tol 1e-6: 2 iterations = 1 step at RMAT scale 23): what a run costs
besides its one propagation step.  Usage: python tools/probe_default_rule.py [--scale 23] (under rocprofv3 --kernel-trace --stats)"""
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
ap.add_argument("--runs", type=int, default=20)
args = ap.parse_args()
pg.load_backend("hip")
adj = rmat_graph(args.scale, 16, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
g = adj.array
deg = np.asarray(pg.degrees(g))
rng = np.random.default_rng(1)
p = np.zeros(g.shape[0])
p[np.sort(rng.choice(np.flatnonzero(deg > 0), 100, replace=False))] = 1.0
sig = pg.to_signal(adj, p)
ranker = pg.PageRank(alpha=0.85)
for _ in range(3):
    ranker.rank(adj, sig)
L.check(L.lib().pgh_sync())
t0 = time.perf_counter()
loop = 0.0
for _ in range(args.runs):
    ranker.rank(adj, sig)
    loop += ranker.last_loop["loop_ms"]
L.check(L.lib().pgh_sync())
dt = (time.perf_counter() - t0) / args.runs
print(f"scale={args.scale} iterations={ranker.last_loop['iterations']} spmv={ranker.last_loop['spmv']} run={dt * 1e6:.0f} us (device loop {loop / args.runs * 1e3:.0f} us) "
      f"-> {g.nnz * ranker.last_loop['spmv'] / dt / 1e9:.1f} GTEPS")
