"""Diagnostic: run the same device loops through two builds of the engine (e.g. tools/variants/libpgh_r01.so and the
current pygrank_amd/csrc/libpgh_hip.so) on a device-generated RMAT graph and compare the results bit for bit, with the
per-kernel HIP-event times of each.  Raw ctypes, one process.
Usage: python tools/ab_compare.py --scale 23 tools/variants/libpgh_r01.so pygrank_amd/csrc/libpgh_hip.so"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pygrank_amd import _lib as L  # noqa: E402

KERNELS = ((0, "spmv"), (6, "pbA"), (7, "pbB"), (1, "fixup"), (5, "combine"), (2, "resid"), (3, "close"))


def bind(path):
    cdll = C.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in L.SIGNATURES.items():
        if hasattr(cdll, name):          # older builds export fewer symbols
            fn = getattr(cdll, name)
            fn.restype, fn.argtypes = restype, argtypes
    return cdll


def run(lib, args):
    assert lib.pgh_init(0) == 0, lib.pgh_last_error()
    g = L.c_graph()
    assert lib.pgh_graph_rmat(args.scale, args.ef, 0.57, 0.19, 0.19, 0, args.norm, args.sym, 0, 0, C.byref(g)) == 0, lib.pgh_last_error()
    vals = [C.c_int64() for _ in range(4)]
    lib.pgh_graph_info(g, *[C.byref(v) for v in vals])
    n, nnz = vals[0].value, vals[2].value
    buf = C.create_string_buffer(1024)
    lib.pgh_graph_format(g, buf, 1024)
    rng = np.random.default_rng(3)
    p = np.zeros(n)
    p[rng.choice(n, 100, replace=False)] = 1.0
    lam = rng.random(n) + 0.5
    vp, vr, vl = L.c_vec(), L.c_vec(), L.c_vec()
    for v in (vp, vr, vl):
        lib.pgh_vec_alloc(n, C.byref(v))
    lib.pgh_vec_h2d_f64(vp, p.ctypes.data, n)
    lib.pgh_vec_h2d_f64(vl, lam.ctypes.data, n)
    out = {"format": buf.value.decode(), "n": n, "nnz": nnz}
    coeffs = np.cumprod(np.r_[1.0, 5.0 / np.arange(3, 40)])
    for name in ("ppr", "ppr_linf_noquot", "absorb", "heat", "cheb"):
        for profile in (0, 1):
            lib.pgh_profile_reset(), lib.pgh_profile_enable(profile)
            res = L.LoopResult()
            if name == "ppr":
                cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=L.ERR_L1, tol=1e-6, max_iters=1000, end_modulo=1, out_scale=1.0,
                                in_norm=100.0, start_from_p=1)
                rc = lib.pgh_ppr_run(g, vp, vr, C.byref(cfg), C.byref(res))
            elif name == "ppr_linf_noquot":
                cfg = L.LoopCfg(alpha=0.85, use_quotient=0, err_kind=L.ERR_LINF, tol=1e-7, max_iters=1000, end_modulo=1, out_scale=2.0,
                                in_norm=100.0, start_from_p=1)
                rc = lib.pgh_ppr_run(g, vp, vr, C.byref(cfg), C.byref(res))
            elif name == "absorb":
                cfg = L.LoopCfg(alpha=0.85, use_quotient=1, err_kind=L.ERR_L1, tol=1e-6, max_iters=1000, end_modulo=1, out_scale=1.0,
                                in_norm=100.0, start_from_p=1)
                rc = lib.pgh_absorb_run(g, vp, vl, vr, C.byref(cfg), C.byref(res))
            else:
                cfg = L.LoopCfg(alpha=0.0, use_quotient=0, err_kind=L.ERR_ITERS, tol=0.0, max_iters=args.terms + 1, end_modulo=1, out_scale=1.0)
                rc = lib.pgh_poly_run(g, vp, coeffs.ctypes.data, len(coeffs), 1 if name == "cheb" else 0, vr, C.byref(cfg), C.byref(res))
            assert rc == 0, lib.pgh_last_error()
            lib.pgh_profile_enable(0)
            if not profile:
                got = np.empty(n, dtype=np.float32)
                lib.pgh_vec_d2h_f32(vr, got.ctypes.data, n)
                out[name] = (got, res.iterations, res.spmv_count, res.loop_ms)
            else:
                parts = []
                for kid, kname in KERNELS:
                    cnt, ms = C.c_int64(), C.c_double()
                    lib.pgh_profile_read(kid, C.byref(cnt), C.byref(ms))
                    if cnt.value:
                        parts.append(f"{kname}={ms.value / cnt.value * 1e3:.1f}")
                out[name + "_kernels"] = " ".join(parts)
    for v in (vp, vr, vl):
        lib.pgh_vec_free(v)
    lib.pgh_graph_destroy(g)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--ef", type=int, default=16)
    ap.add_argument("--norm", type=int, default=0)
    ap.add_argument("--sym", type=int, default=0)
    ap.add_argument("--terms", type=int, default=12)
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    results = []
    for path in args.libs:
        out = run(bind(path), args)
        results.append(out)
        print(f"== {path}: n={out['n']} nnz={out['nnz']}\n   {out['format']}")
        for name in ("ppr", "ppr_linf_noquot", "absorb", "heat", "cheb"):
            got, its, spmv, ms = out[name]
            print(f"   {name:16s} iterations={its:3d} spmv={spmv:3d} loop={ms:8.3f} ms  {ms / max(spmv, 1) * 1e3:7.1f} us/iter  "
                  f"{out['nnz'] * spmv / ms / 1e6:6.1f} GTEPS  sum={float(got.astype(np.float64).sum()):.9g} | {out[name + '_kernels']}", flush=True)
    ok = True
    for other in results[1:]:
        for name in ("ppr", "ppr_linf_noquot", "absorb", "heat", "cheb"):
            a, b = results[0][name], other[name]
            same = np.array_equal(a[0], b[0]) and a[1] == b[1]
            diff = float(np.max(np.abs(a[0].astype(np.float64) - b[0].astype(np.float64))) / max(float(np.max(np.abs(a[0]))), 1e-300))
            print(f"A/B {name:16s}: bit-identical={same} iterations {a[1]} vs {b[1]} rel-Linf={diff:.3e}")
            ok = ok and diff <= 1e-6 and a[1] == b[1]
    print("ab ok" if ok else "ab MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
