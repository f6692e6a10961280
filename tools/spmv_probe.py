"""Perf probe: fixed-iteration PPR on a numpy-generated RMAT graph with per-kernel HIP-event timing.
Usage: python tools/spmv_probe.py --scale 20 --iters 30"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_loops as orc, rmat_np  # noqa: E402  (probe tooling, not product)
import pygrank_amd as pg  # noqa: E402
from pygrank_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    t = time.time()
    A = rmat_np.rmat_csr(args.scale, args.ef, seed=0)
    M = sp.csr_array(orc.normalize(A, "col", True))
    n, nnz = M.shape[0], M.nnz
    print(f"graph scale={args.scale} n={n} nnz={nnz} gen+norm {time.time()-t:.1f}s", flush=True)
    pg.load_backend("hip")
    t = time.time()
    g = pg.scipy_sparse_to_backend(M)
    L.check(L.lib().pgh_sync())
    print(f"upload+transpose {time.time()-t:.2f}s info={g.info()}", flush=True)
    p = np.zeros(n)
    p[rmat_np.seed_nodes(A, 100, seed=1)] = 0.01
    dp, dr = pg.to_array(p), pg.to_array(p)
    bytes_iter = 8 * nnz + 16 * n
    for err_kind, label in ((L.ERR_ITERS, "iters"), (L.ERR_L1, "l1-check")):
        for profile in (0, 1):
            L.check(L.lib().pgh_vec_copy(dr._h, dp._h))
            L.check(L.lib().pgh_profile_reset())
            L.check(L.lib().pgh_profile_enable(profile))
            cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=err_kind, tol=0.0, max_iters=args.iters + 1, end_modulo=1,
                            out_scale=1.0)
            res = L.LoopResult()
            t = time.time()
            L.check(L.lib().pgh_ppr_run(g._h, dp._h, dr._h, C.byref(cfg), C.byref(res)))
            wall = time.time() - t
            L.check(L.lib().pgh_profile_enable(0))
            per = res.loop_ms / max(res.spmv_count, 1)
            print(f"[{label} profile={profile}] spmv={res.spmv_count} loop={res.loop_ms:.3f}ms wall={wall*1e3:.2f}ms "
                  f"per-iter={per*1e3:.1f}us GTEPS={nnz*res.spmv_count/res.loop_ms/1e6:.1f} "
                  f"algGB/s={bytes_iter/per/1e6:.0f}", flush=True)
            if profile:
                for kid, name in enumerate(["spmv", "fixup", "residual", "final"]):
                    cnt, ms = C.c_int64(), C.c_double()
                    L.check(L.lib().pgh_profile_read(kid, C.byref(cnt), C.byref(ms)))
                    if cnt.value:
                        avg = ms.value / cnt.value
                        extra = f" -> {bytes_iter/avg/1e6:.0f} GB/s algorithmic" if name == "spmv" else ""
                        print(f"    {name:9s} launches={cnt.value:4d} avg={avg*1e3:8.1f}us{extra}", flush=True)
    # parity spot check against scipy for the final ranks of a short run
    L.check(L.lib().pgh_vec_copy(dr._h, dp._h))
    cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=L.ERR_L1, tol=1e-6, max_iters=1000, end_modulo=1, out_scale=1.0)
    res = L.LoopResult()
    L.check(L.lib().pgh_ppr_run(g._h, dp._h, dr._h, C.byref(cfg), C.byref(res)))
    t = time.time()
    want, it = orc.pagerank(M, p, alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000, preserve_norm=False)
    cpu = time.time() - t
    got = np.asarray(dr)
    print(f"L1 run: gpu iters={res.iterations} ({res.loop_ms:.2f}ms) cpu iters={it} ({cpu:.2f}s, "
          f"{nnz*(it-1)/cpu/1e9:.2f} GTEPS) rel-Linf={np.max(np.abs(got-want))/np.max(np.abs(want)):.2e}")


if __name__ == "__main__":
    main()
