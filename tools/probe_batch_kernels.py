"""HIP-event time of every kernel of a batched PageRank run (diagnostic): python tools/probe_mm_blocks.py --scale 23 --batch 64"""
import argparse
import ctypes as C
import os
import sys

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
    lib = L.lib()
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
    ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=int(os.environ.get("PROBE_ITERS", "1000")))
    try:
        ranker.propagate(adj, F)
    except Exception as e:
        print("warm-up:", e)
    L.check(lib.pgh_profile_reset())
    L.check(lib.pgh_profile_enable(1))
    try:
        ranker.propagate(adj, F)
    except Exception as e:
        print("timed:", e)
    L.check(lib.pgh_profile_enable(0))
    line = [f"b={args.batch}"]
    for kid, name in ((L.K_SPMM, "partial"), (L.K_FIXUP, "fixup"), (L.K_COMBINE, "combine"), (L.K_RESIDUAL, "residual")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        if cnt.value:
            line.append(f"{name}={ms.value / cnt.value * 1e3:.0f}us x{cnt.value}")
    print(" ".join(line))


if __name__ == "__main__":
    main()
