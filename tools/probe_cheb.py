"""HeatKernel with the reference's "chebyshev" recurrence (f64 route) at the bench scale: per-term time, per-kernel HIP-event
times and the difference to the row-major CSR route (PGH_CHEB_CSR=1).  Usage: python tools/probe_cheb.py --scale 23"""
import argparse
import ctypes as C
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
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--normalization", default="col")
    ap.add_argument("--lib", default=None, help="a diagnostic build of the engine (tools/build_variants.sh) instead of the product library")
    args = ap.parse_args()
    if args.lib:
        L.LIB_PATH = os.path.abspath(args.lib)
    pg.load_backend("hip")
    lib = L.lib()
    adj = rmat_graph(args.scale, 16, seed=0, normalization=args.normalization)
    g = adj.array
    n, nnz = g.shape[0], g.nnz
    deg = np.asarray(pg.degrees(g))
    rng = np.random.default_rng(1)
    p = np.zeros(n)
    p[np.sort(rng.choice(np.flatnonzero(deg > 0), 100, replace=False))] = 1.0
    sig = pg.to_signal(adj, p)
    ranker = pg.HeatKernel(5, coefficient_type="chebyshev", error_type="iters", max_iters=31)
    t0 = time.perf_counter()
    first = np.asarray(ranker.rank(adj, sig).np, dtype=np.float64)
    L.check(lib.pgh_sync())
    print(f"first run (builds the f64 image): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    L.check(lib.pgh_profile_reset())
    L.check(lib.pgh_profile_enable(1))
    ranker.rank(adj, sig)
    L.check(lib.pgh_profile_enable(0))
    parts = []
    for kid, name in ((L.K_SPMV, "partial"), (L.K_FIXUP, "fixup"), (L.K_COMBINE, "combine"), (L.K_PB_GATHER, "pbA"),
                      (L.K_PB_ACCUM, "pbB"), (L.K_FINAL, "close")):
        cnt, ms = C.c_int64(), C.c_double()
        L.check(lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
        if cnt.value:
            parts.append(f"{name}={ms.value / cnt.value * 1e3:.1f}us x{cnt.value}")
    L.check(lib.pgh_sync())
    t0 = time.perf_counter()
    spmv, loop_ms = 0, 0.0
    for _ in range(args.runs):
        ranker.rank(adj, sig)
        spmv += ranker.last_loop["spmv"]
        loop_ms += ranker.last_loop["loop_ms"]
    L.check(lib.pgh_sync())
    dt = time.perf_counter() - t0
    step_us = loop_ms / spmv * 1e3
    print(f"route={'csr' if os.environ.get('PGH_CHEB_CSR') == '1' else 'blocked'} scale={args.scale} nnz={nnz} spmv/run={spmv // args.runs} "
          f"GTEPS={nnz * spmv / dt / 1e9:.1f} device step={step_us:.1f}us = {(8 * nnz + 20 * n) / step_us / 1e3:.0f} GB/s nominal; " + " ".join(parts))
    print(f"checksum: sum={first.sum():.15e} max={first.max():.15e} l1-weighted={np.dot(first, np.arange(n) % 997):.15e}")


if __name__ == "__main__":
    main()
