"""DEV-CONTAINER-ONLY: generate golden vectors by importing the reference (read-only) from
/root/reference.  Output: tests/golden/golden.npz (+ golden_norm.npz).  The reference never ships
to the GPU box; only these data files do.

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ["pygrankBackend"] = "numpy"
os.environ["HOME"] = tempfile.mkdtemp(prefix="pgh_golden_home_")   # import writes ~/.pygrank/config.json
sys.dont_write_bytecode = True
sys.modules["wget"] = types.ModuleType("wget")                      # pygrank/benchmarks/download.py:3
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import pygrank as pg  # noqa: E402

import cases  # noqa: E402

ERR = {"mabs": pg.Mabs, "l1": pg.L1, "linf": pg.MaxDifference, "iters": "iters"}
ALGO = {
    "pagerank": pg.PageRank,
    "heat": pg.HeatKernel,
    "generic": pg.GenericGraphFilter,
    "pagerank_closed": pg.PageRankClosed,
    "absorbing": pg.AbsorbingWalks,
    "lowpass": pg.LowPassRecursiveGraphFilter,
    "sarw": pg.SymmetricAbsorbingRandomWalks,
}


def run_case(graph_cache, name, gkey, algo, kwargs):
    A, directed, p = graph_cache[gkey]
    kwargs = dict(kwargs)
    absorption = kwargs.pop("_absorption", None)
    if "error_type" in kwargs:
        kwargs["error_type"] = ERR[kwargs["error_type"]]
    ranker = ALGO[algo](**kwargs)
    graph = pg.AdjacencyWrapper(A, directed=directed)
    call_kwargs = {}
    if absorption is not None:
        call_kwargs["absorption"] = cases.absorption_vector(absorption, A.shape[0])
    ranks = ranker.rank(graph, p.copy(), **call_kwargs)
    return np.asarray(ranks.np, dtype=np.float64), int(ranker.convergence.iteration)


def main():
    graph_cache = {k: f() for k, f in cases.GRAPHS.items()}
    out = {}
    for name, gkey, algo, kwargs in cases.CASES:
        ranks, iters = run_case(graph_cache, name, gkey, algo, kwargs)
        out[name + "|ranks"] = ranks
        out[name + "|iters"] = np.int64(iters)
        print(f"{name:32s} iters={iters:4d} sum={ranks.sum():.15g} max={ranks.max():.15g}")
    # non-convergence must raise (tests/test_filters.py:59-62)
    A, directed, p = graph_cache["rmat10_dir"]
    try:
        pg.PageRank(max_iters=5, tol=1e-12).rank(pg.AdjacencyWrapper(A, directed=True), p.copy())
        out["rmat10/max_iters_raises"] = np.int64(0)
    except Exception:
        out["rmat10/max_iters_raises"] = np.int64(1)
    # zero personalization -> zero output (tests/test_filters.py:9-10)
    z = pg.PageRank().rank(pg.AdjacencyWrapper(A, directed=True), np.zeros(A.shape[0]))
    out["rmat10/zero_personalization_sum"] = np.float64(pg.sum(z.np))
    # residual measures on a fixed pair (measures/supervised.py:93-106,133-138)
    rng = np.random.default_rng(11)
    u, v = rng.random(1000), rng.random(1000)
    out["residual|u"], out["residual|v"] = u, v
    out["residual|mabs"] = np.float64(pg.Mabs(u)(v))
    out["residual|l1"] = np.float64(pg.L1(u)(v))
    out["residual|linf"] = np.float64(pg.MaxDifference(u)(v))
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)

    # postprocessors and residual-style measures: reference outcomes on a fixed PageRank run
    post = {}
    A, directed, p = graph_cache["rmat10_dir"]
    graph = pg.AdjacencyWrapper(A, directed=True)
    base = lambda: pg.PageRank(0.85, error_type="iters", max_iters=41)       # noqa: E731  (a stopping rule fp32 and fp64 engines share)
    ranks = base().rank(graph, p.copy())
    post["ranks"] = np.asarray(ranks.np, dtype=np.float64)
    for key, algo in (("ordinals", pg.Ordinals(base())), ("top5", pg.Top(base(), 5)), ("top_half", pg.Top(base(), 0.5)),
                      ("threshold", pg.Threshold(base(), 0.02)), ("threshold_inclusive", pg.Threshold(base(), 0.02, inclusive=True)),
                      ("threshold_gap", pg.Threshold(base(), "gap")),
                      ("sweep", pg.Sweep(base())), ("linear_sweep", pg.LinearSweep(base())),
                      ("transformer_exp", pg.Transformer(base())), ("normalize_range", pg.Normalize(base(), "range")),
                      ("normalize_l2", pg.Normalize(base(), "L2"))):
        out = algo.rank(graph, p.copy())
        post["post|" + key] = np.asarray([out[v] for v in range(A.shape[0])], dtype=np.float64)
    rng2 = np.random.default_rng(13)
    u, v = rng2.random(A.shape[0]), rng2.random(A.shape[0])
    post["measure|u"], post["measure|v"] = u, v
    for key, m in (("rmabs", pg.RMabs), ("msq", pg.MSQ), ("msqrt", pg.MSQRT), ("l2", pg.L2), ("euclidean", pg.Euclidean),
                   ("cos", pg.Cos), ("dot", pg.Dot)):
        post["measure|" + key] = np.float64(m(u)(v))
    # AUC (measures/supervised.py:255-263: sklearn roc_curve + auc) of the fixed ranks against binary labels, without ties and with
    # many (scores quantised to 2 decimals of their maximum; zero scores)
    rng3 = np.random.default_rng(17)
    labels = (rng3.random(A.shape[0]) < 0.3).astype(np.float64)
    coarse = np.round(post["ranks"] / post["ranks"].max(), 2)
    post["auc|labels"], post["auc|coarse_scores"] = labels, coarse
    post["auc|ranks"] = np.float64(pg.AUC(labels)(post["ranks"]))
    post["auc|coarse"] = np.float64(pg.AUC(labels)(coarse))
    post["auc|random"] = np.float64(pg.AUC(labels)(u))
    np.savez_compressed(os.path.join(HERE, "golden_post.npz"), **post)

    # normalised CSR fixtures (preprocessing.py:99-142) + degrees (numpy.py:76-77)
    norm = {}
    for gkey in ["rmat10_dir", "weighted300"]:
        A, directed, _ = graph_cache[gkey]
        for normalization in cases.NORMALIZATIONS:
            for renorm in [False, True]:
                M = pg.preprocessor(normalization=normalization, renormalize=renorm)(pg.AdjacencyWrapper(A, directed=directed)).array
                M = M.tocsr()
                M.sort_indices()
                key = f"{gkey}|{normalization}|{int(renorm)}"
                norm[key + "|indptr"] = M.indptr.astype(np.int64)
                norm[key + "|indices"] = M.indices.astype(np.int32)
                norm[key + "|data"] = M.data.astype(np.float64)
                norm[key + "|degrees"] = np.asarray(pg.degrees(M), dtype=np.float64)
                x = np.linspace(0.1, 1.0, A.shape[0])
                norm[key + "|conv"] = np.asarray(pg.conv(x, M), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "golden_norm.npz"), **norm)
    print("wrote", os.path.join(HERE, "golden.npz"), os.path.getsize(os.path.join(HERE, "golden.npz")),
          os.path.getsize(os.path.join(HERE, "golden_norm.npz")))


if __name__ == "__main__":
    main()
