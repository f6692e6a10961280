"""Randomised stress of the propagation kernels on the GPU: many graph shapes / layouts against scipy in fp64.
Usage: python tools/stress_gpu.py --seconds 120 [--seed 0].  Exits non-zero on the first mismatch."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pygrank_amd as pg  # noqa: E402
from pygrank_amd.device import DeviceGraph, DeviceVector  # noqa: E402

EPS = float(np.finfo(np.float32).eps)


def random_graph(rng):
    kind = rng.integers(0, 5)
    n = int(rng.choice([1, 2, 63, 64, 65, 511, 513, 4097, 30000, 70001, 200003, 400000]))
    if kind == 0:                                  # uniform sparse
        deg = float(rng.choice([0.5, 2, 8, 30]))
        nnz = max(1, int(n * deg))
        A = sp.csr_array(sp.coo_array((np.ones(nnz), (rng.integers(0, n, nnz), rng.integers(0, n, nnz))), shape=(n, n)))
    elif kind == 1:                                # power-law columns and rows
        nnz = max(1, int(n * rng.choice([4, 16])))
        r = np.minimum((n * rng.random(nnz) ** 3).astype(np.int64), n - 1)
        c = np.minimum((n * rng.random(nnz) ** 4).astype(np.int64), n - 1)
        A = sp.csr_array(sp.coo_array((np.ones(nnz), (r, c)), shape=(n, n)))
    elif kind == 2:                                # a few hub rows / hub columns plus noise
        nnz = max(1, int(n * 3))
        r, c = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
        hubs = rng.integers(0, n, 3)
        r[: nnz // 3] = hubs[rng.integers(0, 3, nnz // 3)]
        c[nnz // 3: 2 * nnz // 3] = hubs[rng.integers(0, 3, nnz // 3)]
        A = sp.csr_array(sp.coo_array((np.ones(nnz), (r, c)), shape=(n, n)))
    elif kind == 3:                                # banded + empty rows
        offs = [k for k in (0, 1, 5) if k < n]
        A = sp.csr_array(sp.diags([np.ones(n - k) for k in offs], offs, shape=(n, n)))
        mask = sp.diags((rng.random(n) > 0.3).astype(float))
        A = sp.csr_array(mask @ A)
    else:                                          # dense-ish tiny
        n = int(rng.integers(1, 80))
        A = sp.csr_array((rng.random((n, n)) < 0.4).astype(float))
    A.sum_duplicates()
    A.eliminate_zeros()
    if rng.random() < 0.4:                         # real weights -> valued layout
        A.data = A.data * (0.25 + rng.random(A.nnz))
    return A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=-1, help="replay the random sequence but run only this graph on the GPU")
    args = ap.parse_args()
    pg.load_backend("hip")
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    done = 0
    while time.time() < t_end:
        A = random_graph(rng)
        n = A.shape[0]
        for key, val in (("PGH_BLOCKS", str(int(rng.choice([0, 1, 2, 4, 8])))), ("PGH_RELABEL", str(int(rng.integers(0, 2)))),
                         ("PGH_PB", str(int(rng.random() < 0.3))), ("PGH_PB_FORCE", "1"), ("PGH_TRIM", str(int(rng.integers(0, 2)))),
                         ("PGH_PB_HEAVY", str(int(rng.choice([16384, 8, 64])))), ("PGH_PB_HUBMAX", str(int(rng.choice([262144, 150, 4000])))),
                         ("PGH_PB_BINROWS", str(int(rng.choice([4096, 8192, 16384]))))):     # the finish kernel's three shapes
            os.environ[key] = val
        if os.environ["PGH_BLOCKS"] == "0":
            os.environ.pop("PGH_BLOCKS")
        norm = str(rng.choice(["col", "symmetric", "none", "both"]))
        desc = f"#{done} n={n} nnz={A.nnz} norm={norm} env=" + " ".join(f"{k}={os.environ.get(k)}" for k in ("PGH_BLOCKS", "PGH_RELABEL", "PGH_PB", "PGH_TRIM"))
        if os.environ.get("PGH_STRESS_VERBOSE"):
            print(desc, flush=True)
        if args.only >= 0 and done != args.only:                    # same draws as a full run, no GPU work
            rng.integers(0, 3)
            rng.random(n)
            if n >= 2 and A.nnz > 0 and norm in ("col", "symmetric"):
                rng.integers(0, n, min(n, 5))
                rng.integers(0, 2)
            done += 1
            if done > args.only:
                break
            continue
        if args.only >= 0:
            sp.save_npz(os.path.join(ROOT, "gpurun_out", f"stress_graph_{done}.npz"), sp.csr_matrix(A))
        route = int(rng.integers(0, 3))          # device-side normalisation / factored upload / plain valued upload
        if route == 0:
            g = DeviceGraph.from_adjacency(A, norm)
        else:
            from pygrank_amd.preprocessing import normalize_adjacency
            N = normalize_adjacency(A, norm)
            if route == 2:
                N = sp.csr_array((N.data.copy(), N.indices.copy(), N.indptr.copy()), shape=N.shape)   # drops the factors
            g = pg.scipy_sparse_to_backend(N)
        MT = g.download_transposed().astype(np.float64)            # the stored f32 values, exactly
        x = rng.random(n).astype(np.float32).astype(np.float64)
        y = np.asarray(pg.conv(DeviceVector.from_host(x), g))
        ref = MT @ x
        bound = 8 * EPS * (np.abs(MT) @ np.abs(x)) + 1e-30
        if not np.all(np.abs(y - ref) <= bound):
            i = int(np.argmax(np.abs(y - ref) - bound))
            print("MISMATCH conv", desc, g.format(), "row", i, y[i], ref[i], bound[i], flush=True)
            sys.exit(1)
        if n >= 2 and A.nnz > 0 and norm in ("col", "symmetric"):
            from pygrank_amd.preprocessing import Adjacency
            from pygrank_amd.signals import _IdentityMap
            adj = Adjacency(g)
            adj._pygrank_preprocessed = {"hip": adj}
            adj._pygrank_node2id = _IdentityMap(n)
            adj.is_directed = lambda: True
            p = np.zeros(n)
            p[rng.integers(0, n, min(n, 5))] = 1.0
            ranker = pg.PageRank(0.85, error_type="iters", max_iters=8, use_quotient=bool(rng.integers(0, 2)))
            got = np.asarray(ranker.rank(adj, p.copy()).np)
            r = p / p.sum()
            pn = r.copy()
            for _ in range(7):
                r = 0.85 * (MT @ r) + 0.15 * pn
                if ranker.use_quotient:
                    s = r.sum()
                    r = r / s if s != 0 else r
            r = r * p.sum()
            err = np.max(np.abs(got - r)) / max(np.max(np.abs(r)), 1e-30)
            if err > 2e-6:
                print("MISMATCH pagerank", desc, g.format(), "rel", err, flush=True)
                sys.exit(1)
        done += 1
        del g
    print(f"stress ok: {done} graphs in {args.seconds:.0f} s (seed {args.seed})", flush=True)


if __name__ == "__main__":
    main()
