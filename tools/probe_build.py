"""Time-to-first-rank: where a graph build spends its time (pgh_last_build_profile), for the bench's generated graph
and for a scipy upload of the same matrix through the preprocessor, built twice each (the second build finds the allocator's pool warm,
which is what a rank() with assume_immutability=False sees from its second call on: pygrank/core/utils/preprocessing.py:233-287)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def profile():
    from pygrank_amd import _lib as L
    buf = C.create_string_buffer(4096)
    L.check(L.lib().pgh_last_build_profile(buf, 4096))
    return [(k, float(v)) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=23)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--no-upload", action="store_true")
    args = ap.parse_args()
    import pygrank_amd as pg
    from pygrank_amd import _lib as L
    from pygrank_amd.synthetic import rmat_graph
    pg.load_backend("hip")
    lib = L.lib()
    adj = None
    for attempt in range(3):
        adj = None
        L.check(lib.pgh_sync())
        t0 = time.perf_counter()
        adj = rmat_graph(args.scale, args.ef, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
        L.check(lib.pgh_sync())
        dt = time.perf_counter() - t0
        phases = profile()
        print(f"generated graph, build {attempt + 1}: {dt * 1e3:8.1f} ms wall; phases sum {sum(v for _, v in phases):8.1f} ms")
        for k, v in phases:
            print(f"    {v:8.2f} ms  {k}")
    if args.no_upload:
        return
    import scipy.sparse as sp
    MT = adj.array.download_transposed()
    W = sp.csr_array((np.ones(MT.nnz), MT.indices, MT.indptr), shape=MT.shape).T.tocsr()      # the raw adjacency (unit weights)
    W.sort_indices()
    del MT, adj
    sig_p = np.zeros(W.shape[0])
    sig_p[:100] = 1.0
    for attempt in range(3):
        pre = pg.preprocessor(normalization="col", assume_immutability=False)
        graph = pg.AdjacencyWrapper(W, directed=True)
        L.check(lib.pgh_sync())
        t0 = time.perf_counter()
        M = pre(graph)
        L.check(lib.pgh_sync())
        dt = time.perf_counter() - t0
        phases = profile()
        print(f"scipy upload through the preprocessor, build {attempt + 1}: {dt * 1e3:8.1f} ms wall; engine phases sum {sum(v for _, v in phases):8.1f} ms "
              f"({M.array.format()[:60]})")
        for k, v in phases:
            print(f"    {v:8.2f} ms  {k}")
        ranker = pg.PageRank(0.85, preprocessor=pre, error_type=pg.L1, tol=1e-6, max_iters=1000)
        t0 = time.perf_counter()
        out = ranker.rank(graph, sig_p)
        float(out.np[0])
        print(f"    rank() with assume_immutability=False (normalise + upload + image + loop): {(time.perf_counter() - t0) * 1e3:8.1f} ms, "
              f"{ranker.convergence.iteration} iterations")
        del M


if __name__ == "__main__":
    main()
