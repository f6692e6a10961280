"""configs[2] at full size: at which step does every one of the 64 columns stop?  (What per-iteration column compaction -- converged columns
leaving the slab so that a gathered row shrinks below 256 B -- could save: the share of (step, column) pairs that are already done.)"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pygrank_amd as pg
from pygrank_amd.device import DeviceMatrix
from pygrank_amd.synthetic import rmat_graph
pg.load_backend("hip")
adj = rmat_graph(23, 16, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
n = adj.array.shape[0]
cand = np.flatnonzero(np.asarray(pg.degrees(adj.array)) > 0)
feats = DeviceMatrix.empty(n, 64)
for j in range(64):
    col = np.zeros(n)
    col[np.sort(np.random.default_rng(101 + j).choice(cand, 100, replace=False))] = 1.0
    feats.set_column(j, pg.to_signal(adj, col).np)
ranker = pg.PageRank(alpha=0.85, error_type=pg.L1, tol=1e-6, max_iters=1000)
ranker.propagate(adj, feats)
its = [c["iterations"] for c in ranker.last_batches[0]]
hist = collections.Counter(its)
steps = max(its) - 1
live_pairs = sum(i - 1 for i in its)
print("stopping iteration -> columns:", dict(sorted(hist.items())))
print(f"batch steps {steps}; (step, column) pairs that still work: {live_pairs} of {steps * 64} = {100.0 * live_pairs / (steps * 64):.1f} %; "
      f"a perfectly compacted slab would gather {100.0 * (1 - live_pairs / (steps * 64)):.1f} % fewer bytes")
