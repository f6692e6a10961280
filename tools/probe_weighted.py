"""PPR on the bench graph with REAL-VALUED edge weights (`nx ... weight="weight"`,
pygrank/core/utils/preprocessing.py:103): the scale-S RMAT structure with weights rng.random(nnz) + 0.1, "col"
normalisation through the preprocessor (device route: pgh_graph_from_adjacency), so the propagation reads the VALUED
stream (index + f32 value per entry) instead of the value-free one.  Nominal bytes 8 nnz + 16 n per iteration.
Usage: python tools/probe_weighted.py --scale 23 [--parity]"""
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


def weighted_adjacency(scale, ef=16, seed=7):
    """(W, personalization seeds): W = the RMAT adjacency (rows = sources) with real weights, scipy CSR fp64."""
    import scipy.sparse as sp
    adj = rmat_graph(scale, ef, seed=0, normalization="col")
    MT = adj.array.download_transposed()                      # CSR(M^T): structure of A^T
    rng = np.random.default_rng(seed)
    MT = sp.csr_array((rng.random(MT.nnz) + 0.1, MT.indices, MT.indptr), shape=MT.shape)
    W = sp.csr_array(MT.T)                                    # rows = sources again
    W.sort_indices()
    del adj
    return W


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=23)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--parity", action="store_true")
    args = ap.parse_args()
    pg.load_backend("hip")
    t0 = time.time()
    W = weighted_adjacency(args.scale)
    host_s = time.time() - t0
    n, nnz = W.shape[0], W.nnz
    t0 = time.time()
    pre = pg.preprocessor(normalization="col", assume_immutability=True)
    graph = pre(pg.AdjacencyWrapper(W, directed=True))
    L.check(L.lib().pgh_sync())
    up_s = time.time() - t0
    g = graph.array
    outdeg = np.diff(W.indptr)
    rng = np.random.default_rng(1)
    p = np.zeros(n)
    p[np.sort(rng.choice(np.flatnonzero(outdeg > 0), 100, replace=False))] = 1.0
    sig = pg.to_signal(graph, p)
    rows = []
    for name, ranker in (("PageRank a=0.85 L1<=1e-6", pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000, preprocessor=pre)),
                         ("PageRank a=0.85, 50 iterations", pg.PageRank(0.85, error_type="iters", max_iters=51, preprocessor=pre))):
        ranks = ranker.rank(graph, sig)
        L.check(L.lib().pgh_sync())
        t0 = time.perf_counter()
        spmv, loop_ms = 0, 0.0
        for _ in range(args.runs):
            ranks = ranker.rank(graph, sig)
            spmv += ranker.last_loop["spmv"]
            loop_ms += ranker.last_loop["loop_ms"]
        L.check(L.lib().pgh_sync())
        dt = time.perf_counter() - t0
        step_us = loop_ms / spmv * 1e3
        rows.append(f"{name:34s} n={n} nnz={nnz} iterations={ranker.last_loop['iterations']:3d} GTEPS={nnz * spmv / dt / 1e9:6.1f} "
                    f"device step={step_us:6.1f} us = {(8 * nnz + 16 * n) / step_us / 1e3:6.0f} GB/s nominal "
                    f"({(8 * nnz + 16 * n) / step_us / 1e3 / 8000:.3f} of 8 TB/s)  format={g.format()}")
        if args.parity and "L1" in name:
            from oracle import ref_loops as orc
            t0 = time.time()
            want, want_iters = orc.pagerank(orc.normalize(W, "col", True), p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000)
            got = np.asarray(ranks.np, dtype=np.float64)
            rows.append(f"   parity vs the oracle: rel-Linf={np.max(np.abs(got - want)) / np.max(np.abs(want)):.3e}, iterations "
                        f"{ranker.last_loop['iterations']} / {want_iters} (oracle {time.time() - t0:.0f} s)")
    print(f"host adjacency {host_s:.1f} s, preprocess + upload {up_s:.1f} s")
    print("\n".join(rows))


if __name__ == "__main__":
    main()
