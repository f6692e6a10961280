"""rank(..., graph_dropout=r) at the bench scale: the device loop on the blocked layouts (pgh_ppr_run_dropout) against the same loop on
the row-major CSR kernel (PGH_DROPOUT_CSR=1) and against the hook protocol (one engine call per backend primitive, round 3).
Usage: python tools/probe_dropout.py [--scale 23]"""
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
ap.add_argument("--rate", type=float, default=0.3)
ap.add_argument("--iters", type=int, default=21)
ap.add_argument("--hooks", action="store_true", help="the hook protocol instead of the device loop")
args = ap.parse_args()
pg.load_backend("hip")
adj = rmat_graph(args.scale, 16, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
g = adj.array
deg = np.asarray(pg.degrees(g))
rng = np.random.default_rng(1)
p = np.zeros(g.shape[0])
p[np.sort(rng.choice(np.flatnonzero(deg > 0), 100, replace=False))] = 1.0
sig = pg.to_signal(adj, p)
ranker = pg.PageRank(alpha=0.85, error_type="iters", max_iters=args.iters)
ranker.fused_dropout = not args.hooks
t0 = time.perf_counter()
ranker.rank(adj, sig, graph_dropout=args.rate)
L.check(L.lib().pgh_sync())
print(f"first run (builds the index words): {(time.perf_counter() - t0) * 1e3:.1f} ms")
runs = 3
pg.backend.hip.set_dropout_seed(5)
t0 = time.perf_counter()
for _ in range(runs):
    out = ranker.rank(adj, sig, graph_dropout=args.rate)
L.check(L.lib().pgh_sync())
dt = (time.perf_counter() - t0) / runs
steps = args.iters - 1
route = "hook protocol" if args.hooks else ("device loop, row-major kernel" if os.environ.get("PGH_DROPOUT_CSR") == "1" else "device loop, blocked layouts")
print(f"{route}: scale={args.scale} rate={args.rate} {steps} steps per run: {dt * 1e3:.2f} ms per run = {dt / steps * 1e6:.0f} us per step "
      f"-> {g.nnz * steps / dt / 1e9:.1f} GTEPS; checksum {float(np.asarray(out.np, dtype=np.float64).sum()):.6f}")
