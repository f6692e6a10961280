"""The backend-primitive route (what the unmodified reference filters execute through backend/hip.py: one engine call per conv / sum /
abs / - / * ...; pygrank/core/backend/__init__.py:59-80) against the fused device loop on the bench graph: GTEPS, iterations, and the
difference of the results; with lazy vectors (device.LazyVector: resident iterates, one engine step per formula) and without."""
import argparse
import sys
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=23)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--runs", type=int, default=3)
    args = ap.parse_args()
    import pygrank_amd as pg
    from pygrank_amd import _lib as L, device
    from pygrank_amd.synthetic import rmat_graph
    pg.load_backend("hip")
    adj = rmat_graph(args.scale, args.ef, seed=0, normalization="col", a=0.57, b=0.19, c=0.19)
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    cand = np.flatnonzero(np.asarray(pg.degrees(g)) > 0)
    p = np.zeros(n)
    p[np.sort(np.random.default_rng(1).choice(cand, 100, replace=False))] = 1.0
    sig = pg.to_signal(adj, p)
    makers = {
        "ppr_l1_1e-6": lambda: pg.PageRank(alpha=0.85, error_type=pg.L1, tol=1e-6, max_iters=1000),
        "ppr_mabs_default": lambda: pg.PageRank(alpha=0.85, tol=1e-6, max_iters=1000),
        "heat_kernel_t5_31": lambda: pg.HeatKernel(5, error_type="iters", max_iters=31),
        "absorbing_a085_l1": lambda: pg.AbsorbingWalks(0.85, error_type=pg.L1, tol=1e-6, max_iters=1000),
    }
    for name, make in makers.items():
        fused = make()
        want = np.asarray(fused.rank(adj, sig).np)
        L.check(L.lib().pgh_sync())
        t0 = time.perf_counter()
        for _ in range(args.runs):
            fused.rank(adj, sig)
        L.check(L.lib().pgh_sync())
        t_fused = (time.perf_counter() - t0) / args.runs
        it_fused = fused.convergence.iteration
        print(f"{name:22s} fused      : {it_fused:3d} iterations {t_fused * 1e3:8.2f} ms/run {nnz * (it_fused - 1) / t_fused / 1e9:7.1f} GTEPS")
        for lazy in (True, False):
            device.LAZY = lazy
            generic = make()
            generic._fused_loop = lambda *a, **k: False
            generic._fused_rank = lambda *a, **k: None
            got = np.asarray(generic.rank(adj, sig).np)
            L.check(L.lib().pgh_sync())
            t0 = time.perf_counter()
            for _ in range(args.runs):
                out = generic.rank(adj, sig)
                np.asarray(out.np[0])                   # somebody looks at the result
            L.check(L.lib().pgh_sync())
            t_gen = (time.perf_counter() - t0) / args.runs
            it = generic.convergence.iteration
            err = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
            print(f"{name:22s} primitives {'lazy ' if lazy else 'eager'}: {it:3d} iterations {t_gen * 1e3:8.2f} ms/run "
                  f"{nnz * (it - 1) / t_gen / 1e9:7.1f} GTEPS  ({t_gen / max(it - 1, 1) * 1e6:6.1f} us per iteration)  rel-Linf vs fused {err:.2e}")
        device.LAZY = True


if __name__ == "__main__":
    main()
