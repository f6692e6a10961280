"""Multi-seed workload (BASELINE.json configs[2]): batch of b personalization vectors on RMAT, batched PageRank vs
the same seeds run one by one.  Usage: python tools/probe_batch.py --scale 23 --batch 64"""
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
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    pg.load_backend("hip")
    adj = rmat_graph(args.scale, 16, seed=0)
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    deg = np.asarray(pg.degrees(g))
    cand = np.flatnonzero(deg > 0)
    feats = np.zeros((n, args.batch))
    for j in range(args.batch):
        rng = np.random.default_rng(1 + j)
        feats[np.sort(rng.choice(cand, 100, replace=False)), j] = 1.0
    F = pg.to_primitive(feats)
    ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000)
    out = None
    for rep in range(2):
        del out                                   # (a result kept across the call costs the next one a 2 GB hipMalloc: 0-300 ms of wall)
        L.check(L.lib().pgh_sync())
        t0 = time.perf_counter()
        out = ranker.propagate(adj, F)
        L.check(L.lib().pgh_sync())
        dt = time.perf_counter() - t0
    info = ranker.last_batches[0]
    spmv = sum(c["spmv"] for c in info)
    steps = max(c["spmv"] for c in info)
    print(f"batched: b={args.batch} wall={dt*1e3:.1f}ms device-loop={info[0]['loop_ms']:.1f}ms batch-steps={steps} "
          f"edge-vector products/s={nnz*spmv/dt/1e9:.1f} G ({nnz*spmv/(info[0]['loop_ms']*1e-3)/1e9:.1f} G in the device loop), "
          f"per batch step {info[0]['loop_ms']/steps*1e3:.0f}us")
    cols = np.asarray(out)
    t0 = time.perf_counter()
    single_spmv = 0
    worst = 0.0
    for j in range(min(args.batch, 8)):
        r = ranker.rank(adj, feats[:, j])
        single_spmv += ranker.last_loop["spmv"]
        ref = np.asarray(r.np)
        worst = max(worst, float(np.max(np.abs(cols[:, j] - ref)) / np.max(np.abs(ref))))
        if ranker.last_loop["iterations"] != info[j]["iterations"]:
            print(f"  column {j}: batched stopped at {info[j]['iterations']}, single at {ranker.last_loop['iterations']} "
                  f"(single residual {ranker.last_loop['last_error']:.3e} vs tol 1e-6)")
    dt1 = time.perf_counter() - t0
    print(f"one by one (8 seeds): {dt1/8*1e3:.2f}ms per seed -> {nnz*single_spmv/dt1/1e9:.1f} G edge-vector products/s; "
          f"batched vs single rel-Linf={worst:.2e}")


if __name__ == "__main__":
    main()
