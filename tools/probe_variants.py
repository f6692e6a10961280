"""Diagnostic: time the PPR loop of several engine builds (tools/variants/*.so) on a device-generated RMAT graph.
Raw ctypes on purpose (one process, several libraries).  Usage: python tools/probe_variants.py --scale 22 lib1.so lib2.so"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygrank_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--copy", action="store_true")
    ap.add_argument("--seeds", type=int, default=0, help="personalization = this many seed nodes with out-edges (the bench's shape); 0 = dense 1/n")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    for path in args.libs:
        lib = L._bind(C.CDLL(os.path.abspath(path)))
        assert lib.pgh_init(0) == 0, lib.pgh_last_error()
        g = L.c_graph()
        assert lib.pgh_graph_rmat(args.scale, args.ef, 0.57, 0.19, 0.19, 0, 0, 0, 0, 0, C.byref(g)) == 0, lib.pgh_last_error()
        vals = [C.c_int64() for _ in range(4)]
        lib.pgh_graph_info(g, *[C.byref(v) for v in vals])
        n, nnz = vals[0].value, vals[2].value
        vp, vr = L.c_vec(), L.c_vec()
        lib.pgh_vec_alloc(n, C.byref(vp)), lib.pgh_vec_alloc(n, C.byref(vr))
        lib.pgh_vec_fill(vp, 1.0 / n)
        if args.seeds > 0:
            import numpy as np
            lib.pgh_graph_degrees(g, vr)
            deg = np.empty(n, dtype=np.float32)
            lib.pgh_vec_d2h_f32(vr, deg.ctypes.data_as(C.c_void_p), n)
            cand = np.flatnonzero(deg > 0)
            pick = np.sort(np.random.default_rng(1).choice(cand, size=args.seeds, replace=False))
            p = np.zeros(n, dtype=np.float32)
            p[pick] = 1.0 / args.seeds
            lib.pgh_vec_h2d_f32(vp, p.ctypes.data_as(C.c_void_p), n)
        bytes_iter = 8 * nnz + 16 * n
        out = [f"{os.path.basename(path):28s} n={n} nnz={nnz}"]
        for profile in (0, 1):
            lib.pgh_vec_copy(vr, vp)
            lib.pgh_profile_reset(), lib.pgh_profile_enable(profile)
            cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=L.ERR_L1, tol=0.0, max_iters=args.iters + 1, end_modulo=1, out_scale=1.0)
            res = L.LoopResult()
            assert lib.pgh_ppr_run(g, vp, vr, C.byref(cfg), C.byref(res)) == 0, lib.pgh_last_error()
            lib.pgh_profile_enable(0)
            per = res.loop_ms / max(res.spmv_count, 1)
            if not profile:
                out.append(f"loop/iter={per*1e3:7.1f}us GTEPS={nnz/per/1e6:6.1f} alg={bytes_iter/per/1e6:5.0f}GB/s")
            else:
                parts = []
                for kid, name in ((0, "spmv"), (6, "pbA"), (7, "pbB"), (1, "fixup"), (5, "combine"), (2, "resid"), (3, "final")):
                    cnt, ms = C.c_int64(), C.c_double()
                    lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms))
                    if cnt.value:
                        parts.append(f"{name}={ms.value/cnt.value*1e3:.1f}us")
                out.append(" ".join(parts))
        print(" | ".join(out), flush=True)
        if args.copy:
            m = 1 << 27
            va, vb, vc = L.c_vec(), L.c_vec(), L.c_vec()
            for v in (va, vb, vc):
                lib.pgh_vec_alloc(m, C.byref(v)), lib.pgh_vec_fill(v, 1.0)
            t = L.c_timer()
            lib.pgh_timer_create(C.byref(t))
            for _ in range(3):
                lib.pgh_axpby(0.5, va, 0.5, vb, vc)
            lib.pgh_timer_start(t)
            for _ in range(10):
                lib.pgh_axpby(0.5, va, 0.5, vb, vc)
            lib.pgh_timer_stop(t)
            ms = C.c_double()
            lib.pgh_timer_elapsed_ms(t, C.byref(ms))
            print(f"    stream axpby (2 reads + 1 write of {m*4/2**20:.0f} MiB): {3*4*m*10/ms.value/1e6:.0f} GB/s", flush=True)
            lib.pgh_timer_start(t)
            for _ in range(10):
                lib.pgh_vec_copy(vc, va)
            lib.pgh_timer_stop(t)
            lib.pgh_timer_elapsed_ms(t, C.byref(ms))
            print(f"    hipMemcpy D2D: {2*4*m*10/ms.value/1e6:.0f} GB/s", flush=True)
            for v in (va, vb, vc):
                lib.pgh_vec_free(v)
        lib.pgh_vec_free(vp), lib.pgh_vec_free(vr), lib.pgh_graph_destroy(g)
        lib.pgh_shutdown()


if __name__ == "__main__":
    main()
