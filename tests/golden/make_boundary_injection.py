"""DEV-CONTAINER-ONLY: proof of the drop-in boundary ("verification trick").

The backend module of this build (pygrank_amd/backend/hip.py, the 29 functions of pygrank/core/backend/specification.py)
is registered in the UNMODIFIED reference loader (pygrank/core/backend/__init__.py:40-84) in the slot of an allow-listed
engine that is not installed here ("matvec"); `pygrank.load_backend("matvec")` then makes the reference's own
PageRank / HeatKernel / AbsorbingWalks / SymmetricAbsorbingRandomWalks code drive this build's primitives: conv, sum,
degrees, the DeviceVector operator protocol -- one engine call per backend call, exactly what a maintainer gets from
INTEGRATION.md section A.  No GPU exists in this container, so the C-ABI underneath is the host restatement of
oracle/host_abi.cpp (same ABI, f32 storage); on an MI355X the same module binds libpgh_hip.so.

Each case runs twice through the reference's filters: on the reference's numpy backend and on the injected module; the
iteration counts and the relative L-inf difference go to tests/golden/boundary_injection.json, which
tests/test_boundary_injection.py holds to equal iterations and <= 1e-6.

Run:  python tests/golden/make_boundary_injection.py
"""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, os.path.join(ROOT, "tests"), ROOT]
os.environ["pygrankBackend"] = "numpy"
os.environ["HOME"] = tempfile.mkdtemp(prefix="pgh_boundary_home_")   # import writes ~/.pygrank/config.json
sys.dont_write_bytecode = True
sys.modules["wget"] = types.ModuleType("wget")                        # pygrank/benchmarks/download.py:3
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import pygrank as ref  # noqa: E402  (the reference, unmodified)

import cases  # noqa: E402
import host_double  # noqa: E402

SLOT = "matvec"          # allow-listed by the reference loader (backend/__init__.py:41), not installed in this image


def injected_module():
    host_double.install()
    import pygrank_amd.backend.hip as hip
    mod = types.ModuleType("pygrank.core.backend." + SLOT)
    for name in dir(hip):
        if not name.startswith("__"):
            setattr(mod, name, getattr(hip, name))
    mod.backend_name = lambda: SLOT      # Backend.__exit__ restores the previous engine by name (backend/__init__.py:30-37)
    return mod


ERR = {"mabs": ref.Mabs, "l1": ref.L1, "linf": ref.MaxDifference, "iters": "iters"}
ALGO = {"pagerank": ref.PageRank, "heat": ref.HeatKernel, "absorbing": ref.AbsorbingWalks, "generic": ref.GenericGraphFilter,
        "pagerank_closed": ref.PageRankClosed, "sarw": ref.SymmetricAbsorbingRandomWalks}
# golden cases whose tolerance an fp32 engine can honour (tol >= eps(fp32) or fixed iteration counts; the reference clamps
# tol to max(tol, backend.epsilon()), convergence.py:101) and whose recurrence does not amplify fp32 rounding (the
# chebyshev cases: this route evaluates them with per-step fp32 primitives, see tests/parity_common.py)
CASES = ["er10k/pagerank_default", "er10k/pagerank_l1", "er10k/heat_taylor", "er10k/heat_default", "er10k/absorbing_085",
         "er10k/sarw_default", "rmat10/pagerank_default", "rmat10/pagerank_l1", "rmat10/pagerank_iters", "rmat10/heat_taylor",
         "rmat10/absorbing_085", "rmat10/sarw_l1", "rmat12/pagerank_default", "rmat12/heat_taylor", "w300/pagerank_default",
         "w300/heat_taylor"]


def run(name, graphs):
    _, gkey, algo, kwargs = next(c for c in cases.CASES if c[0] == name)
    A, directed, p = graphs[gkey]
    kwargs = dict(kwargs)
    kwargs.pop("_absorption", None)
    if "error_type" in kwargs:
        kwargs["error_type"] = ERR[kwargs["error_type"]]
    ranker = ALGO[algo](**kwargs)
    ranks = ranker.rank(ref.AdjacencyWrapper(A, directed=directed), p.copy())
    return np.asarray(ref.to_array(ranks.np) if ref.backend_name() == "numpy" else np.asarray(ranks.np), dtype=np.float64), \
        int(ranker.convergence.iteration)


def main():
    graphs = {k: f() for k, f in cases.GRAPHS.items()}
    ref.load_backend("numpy")
    want = {name: run(name, graphs) for name in CASES}
    import pygrank.core.backend as loader
    loader._imported_mods[SLOT] = injected_module()
    ref.load_backend(SLOT)
    assert ref.backend_name() == SLOT
    out = {"_meta": {"slot": SLOT, "reference": "pygrank " + str(getattr(ref, "__version__", "0.2.12")) + " (unmodified, /root/reference)",
                     "engine_under_the_module": "oracle/host_abi.cpp (no GPU in the dev container)"}}
    for name in CASES:
        got, iters = run(name, graphs)
        w, w_iters = want[name]
        out[name] = {"iterations_numpy": w_iters, "iterations_injected": iters,
                     "rel_linf": float(np.max(np.abs(got - w)) / np.max(np.abs(w)))}
        print(f"{name:28s} iterations {w_iters:3d} / {iters:3d}   rel-Linf {out[name]['rel_linf']:.3e}")
    ref.load_backend("numpy")
    with open(os.path.join(HERE, "boundary_injection.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "boundary_injection.json"))


if __name__ == "__main__":
    main()
