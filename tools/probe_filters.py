"""The other hot-path workloads of SURVEY.md 8d at the bench scale: HeatKernel (30 polynomial terms, configs[3]) and
AbsorbingWalks, with the per-step algorithmic-byte rates next to PageRank.  Usage: python tools/probe_filters.py --scale 23"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=23)
    ap.add_argument("--runs", type=int, default=5)
    args = ap.parse_args()
    pg.load_backend("hip")
    adj = rmat_graph(args.scale, 16, seed=0, normalization="col")
    sym = rmat_graph(args.scale, 16, seed=0, normalization="symmetric", symmetrize=True)
    rows = []
    for name, graph, ranker, bytes_per_step in (
            ("PageRank a=0.85 L1<=1e-6 (col)", adj, pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000), lambda nnz, n: 8 * nnz + 16 * n),
            ("HeatKernel t=5, 31 iterations (col)", adj, pg.HeatKernel(5, error_type="iters", max_iters=31), lambda nnz, n: 8 * nnz + 20 * n),
            ("HeatKernel t=5 chebyshev, 31 iterations", adj, pg.HeatKernel(5, coefficient_type="chebyshev", error_type="iters", max_iters=31),
             lambda nnz, n: 8 * nnz + 20 * n),
            ("AbsorbingWalks a=0.85 L1<=1e-6 (col)", adj, pg.AbsorbingWalks(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000), lambda nnz, n: 8 * nnz + 24 * n),
            ("PageRank a=0.85 L1<=1e-6 (A+A^T, symmetric)", sym, pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000),
             lambda nnz, n: 8 * nnz + 16 * n)):
        g = graph.array
        n, nnz = g.shape[0], g.nnz
        deg = np.asarray(pg.degrees(g))
        rng = np.random.default_rng(1)
        p = np.zeros(n)
        p[np.sort(rng.choice(np.flatnonzero(deg > 0), 100, replace=False))] = 1.0
        sig = pg.to_signal(graph, p)
        ranker.rank(graph, sig)
        L.check(L.lib().pgh_sync())
        t0 = time.perf_counter()
        spmv, loop_ms = 0, 0.0
        for _ in range(args.runs):
            ranker.rank(graph, sig)
            spmv += ranker.last_loop["spmv"]
            loop_ms += ranker.last_loop["loop_ms"]
        L.check(L.lib().pgh_sync())
        dt = time.perf_counter() - t0
        per_step_us = loop_ms / spmv * 1e3
        rows.append(f"{name:48s} nnz={nnz} iterations={ranker.last_loop['iterations']:3d} spmv/run={spmv // args.runs:3d} "
                    f"run={dt / args.runs * 1e3:7.2f} ms  GTEPS={nnz * spmv / dt / 1e9:6.1f}  device step={per_step_us:6.1f} us "
                    f"= {bytes_per_step(nnz, n) / per_step_us / 1e3:6.0f} GB/s nominal ({g.format().split(',')[0]})")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
