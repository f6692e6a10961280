"""Why do the columns of propagate() stop a step away from the oracle ten times as often as single-vector runs?
Random graphs / feature columns as tests/stress_filters.py draws them; per column: the oracle's stopping iteration, the batch loop's
(pgh_ppr_run_batch), the single-vector loop's (pgh_ppr_run on the same column).  Histograms of (engine - oracle), and of how close the
oracle's residual at its stopping step was to the tolerance for the columns that differ."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pygrank_amd as pg  # noqa: E402
from oracle import ref_loops as orc  # noqa: E402
from pygrank_amd.device import DeviceGraph  # noqa: E402
from pygrank_amd.preprocessing import Adjacency  # noqa: E402
from pygrank_amd.signals import _IdentityMap  # noqa: E402
from stress_gpu import random_graph  # noqa: E402

EPS32 = float(np.finfo(np.float32).eps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    pg.load_backend("hip")
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    hist = dict(batch={}, single={}, single_f64={})
    margins = dict(batch=[], single=[], single_f64=[])
    columns = 0
    while time.time() < t_end:
        A = random_graph(rng)
        n = A.shape[0]
        if n < 2 or A.nnz == 0 or n > 80000:
            continue
        norm = str(rng.choice(["col", "symmetric"]))
        W = sp.csr_array(A)
        g = DeviceGraph.from_adjacency(W, norm)
        adj = Adjacency(g)
        adj._pygrank_node2id = _IdentityMap(n)
        adj._pygrank_preprocessed = {"hip": adj}
        M = sp.csr_array(g.download_transposed().T.astype(np.float64))
        b = int(rng.choice([1, 3, 12, 17, 33]))
        feats = np.zeros((n, b))
        for j in range(b):
            feats[rng.integers(0, n, 3), j] = 1.0 + j
        kw = dict(alpha=0.85, error_type="l1", tol=1e-6, max_iters=300)
        try:
            oracle = [orc.pagerank(M, feats[:, j], eps=EPS32, **kw) for j in range(b)]
            ranker = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=300)
            ranker.propagate(adj, pg.to_primitive(feats))
            batch_its = [c["iterations"] for c in ranker.last_batches[0]]
            single_its, f64_its = [], []
            for j in range(b):
                one = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=300)
                one.rank(adj, feats[:, j].copy())
                single_its.append(one.convergence.iteration)
                # ... and with f64 iterates (dtype="float64"): the trajectory of the fp64 reference, not an f32 one 1e-7 beside it
                if oracle[j][1] in (batch_its[j], single_its[j]) and batch_its[j] == single_its[j]:
                    f64_its.append(None)
                    continue
                two = pg.PageRank(0.85, error_type=pg.L1, tol=1e-6, max_iters=300, dtype="float64")
                two.rank(adj, feats[:, j].copy())
                f64_its.append(two.convergence.iteration)
        except Exception:
            continue
        for j in range(b):
            it = oracle[j][1]
            oracle64 = None
            for name, its in (("batch", batch_its[j]), ("single", single_its[j]), ("single_f64", f64_its[j])):
                if its is None:
                    continue
                if name == "single_f64":                   # (its tolerance is not clamped at fp32 eps; 1e-6 lies above both)
                    oracle64 = orc.pagerank(M, feats[:, j], **kw)[1]
                    it = oracle64
                d = its - it
                hist[name][d] = hist[name].get(d, 0) + 1
                if d != 0:
                    # the oracle's residual at its stopping check, relative to the tolerance (how close the call was)
                    prev = orc.pagerank(M, feats[:, j], alpha=0.85, error_type="iters", max_iters=it - 1, eps=EPS32)[0]
                    last = orc.pagerank(M, feats[:, j], alpha=0.85, error_type="iters", max_iters=it, eps=EPS32)[0]
                    s = np.abs(feats[:, j]).sum()
                    margins[name].append(float(np.abs(last - prev).sum() / s / 1e-6))
            columns += 1
    print(f"{columns} columns (seed {args.seed}); env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("PGH_")))
    for name in ("batch", "single", "single_f64"):
        total = sum(hist[name].values())
        off = total - hist[name].get(0, 0)
        print(f"  {name:10s} iterations engine - oracle: " + " ".join(f"{k:+d}:{v}" for k, v in sorted(hist[name].items())) + f"   ({100.0 * off / max(total, 1):.2f} % off)")
        if margins[name]:
            m = np.array(margins[name])
            print(f"         oracle residual / tol at its stopping check, columns that differ: median {np.median(m):.4f}, 10-90 % [{np.quantile(m, 0.1):.4f}, {np.quantile(m, 0.9):.4f}]")


if __name__ == "__main__":
    main()
