"""Per-run latency of HeatKernel(t=5, taylor).rank on a small graph: python tools/probe_latency_heat.py SCALE"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygrank_amd as pg
from pygrank_amd import _lib as L
from pygrank_amd.synthetic import rmat_graph

pg.load_backend("hip")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 12
adj = rmat_graph(scale, 16, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
n = adj.array.shape[0]
p = np.zeros(n); p[:100] = 1.0
sig = pg.to_signal(adj, p)
ranker = pg.HeatKernel(5, error_type=pg.L1, tol=1e-9, max_iters=100)
for _ in range(5):
    ranker.rank(adj, sig)
L.check(L.lib().pgh_sync())
N = 200
t0 = time.perf_counter()
for _ in range(N):
    ranker.rank(adj, sig)
L.check(L.lib().pgh_sync())
dt = (time.perf_counter() - t0) / N
print(f"scale={scale} HeatKernel rank() latency {dt*1e6:.0f} us, device loop {ranker.last_loop['loop_ms']*1e3:.0f} us, iterations {ranker.last_loop['iterations']}")
