"""Declarative list of golden cases shared by make_golden.py (reference-driven, dev container only),
tests/test_oracle_golden.py (oracle vs fixtures, CPU) and tests/test_gpu_parity.py (HIP vs fixtures).

A case = (graph key, algorithm key, constructor kwargs).  Graph builders only use numpy / scipy /
networkx (present on the GPU box) -- never /root/reference.
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import rmat_np  # noqa: E402


def graph_er10k():
    """cfg1 of BASELINE.json: 10k nodes / 80k undirected edges, seeds {0,1,2}."""
    import networkx as nx
    G = nx.gnm_random_graph(10000, 80000, seed=0)
    A = sp.csr_array(nx.to_scipy_sparse_array(G, dtype=float))
    p = np.zeros(10000)
    p[[0, 1, 2]] = 1.0
    return A, False, p


def graph_rmat10_dir():
    """directed RMAT scale 10, edge factor 8: duplicate-edge weights, self loops, dangling rows."""
    A = rmat_np.rmat_csr(10, 8, seed=0)
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 20, seed=1)] = 1.0
    return A, True, p


def graph_rmat12_sym():
    """undirected (A + A^T) RMAT scale 12, edge factor 8."""
    A = rmat_np.rmat_csr(12, 8, seed=3)
    A = sp.csr_array(A + A.T)
    A.sort_indices()
    p = np.zeros(A.shape[0])
    p[rmat_np.seed_nodes(A, 50, seed=2)] = 1.0
    return A, False, p


def graph_weighted300():
    """300-node directed graph, random fp weights, 10% dangling rows, non-uniform personalization."""
    rng = np.random.default_rng(7)
    n = 300
    rows = rng.integers(0, n, 3000)
    cols = rng.integers(0, n, 3000)
    keep = rows % 10 != 3                   # rows with id % 10 == 3 have no out-edges
    w = rng.random(3000) + 0.1
    A = sp.coo_array((w[keep], (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    p = np.zeros(n)
    idx = rng.choice(n, 10, replace=False)
    p[idx] = rng.random(10) * 3 + 0.5
    return A, True, p


GRAPHS = {
    "er10k": graph_er10k,
    "rmat10_dir": graph_rmat10_dir,
    "rmat12_sym": graph_rmat12_sym,
    "weighted300": graph_weighted300,
}

_HK = dict(t=5, error_type="iters", max_iters=31)
_W20 = [0.85 ** i for i in range(20)]

# (case name, graph, algo, kwargs)
CASES = [
    # ---- cfg1 (SURVEY 8c golden scalars are re-checked in test_oracle_golden.py) ----
    ("er10k/pagerank_default", "er10k", "pagerank", dict(alpha=0.85)),
    ("er10k/pagerank_tol1e-9", "er10k", "pagerank", dict(alpha=0.85, tol=1e-9, max_iters=1000)),
    ("er10k/pagerank_noquot", "er10k", "pagerank", dict(alpha=0.85, use_quotient=False, tol=1e-12, max_iters=1000)),
    ("er10k/pagerank_col", "er10k", "pagerank", dict(alpha=0.85, normalization="col", tol=1e-9, max_iters=1000)),
    ("er10k/pagerank_l1", "er10k", "pagerank", dict(alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000)),
    ("er10k/heat_taylor", "er10k", "heat", dict(**_HK)),
    ("er10k/heat_cheb", "er10k", "heat", dict(coefficient_type="chebyshev", **_HK)),
    ("er10k/heat_default", "er10k", "heat", dict(t=5)),
    ("er10k/absorbing_085", "er10k", "absorbing", dict(alpha=0.85, max_iters=1000)),
    ("er10k/absorbing_default", "er10k", "absorbing", dict(tol=1e-9, max_iters=1000)),
    # ---- directed RMAT with dangling rows / duplicate-edge weights ----
    ("rmat10/pagerank_default", "rmat10_dir", "pagerank", dict(alpha=0.85)),
    ("rmat10/pagerank_l1", "rmat10_dir", "pagerank", dict(alpha=0.85, error_type="l1", tol=1e-6, max_iters=1000)),
    ("rmat10/pagerank_linf", "rmat10_dir", "pagerank", dict(alpha=0.9, error_type="linf", tol=1e-8, max_iters=1000)),
    ("rmat10/pagerank_iters", "rmat10_dir", "pagerank", dict(alpha=0.85, error_type="iters", max_iters=51)),
    ("rmat10/pagerank_noquot", "rmat10_dir", "pagerank", dict(alpha=0.85, use_quotient=False, tol=1e-10, max_iters=1000)),
    ("rmat10/pagerank_sym", "rmat10_dir", "pagerank", dict(alpha=0.85, normalization="symmetric", tol=1e-9, max_iters=1000)),
    ("rmat10/pagerank_renorm", "rmat10_dir", "pagerank", dict(alpha=0.85, renormalize=True, tol=1e-9, max_iters=1000)),
    ("rmat10/pagerank_modulo", "rmat10_dir", "pagerank", dict(alpha=0.85, tol=1e-9, end_modulo=4, max_iters=1000)),
    ("rmat10/heat_taylor", "rmat10_dir", "heat", dict(**_HK)),
    ("rmat10/heat_cheb", "rmat10_dir", "heat", dict(coefficient_type="chebyshev", **_HK)),
    ("rmat10/generic20", "rmat10_dir", "generic", dict(weights=_W20, tol=1e-12)),
    ("rmat10/generic_cheb", "rmat10_dir", "generic", dict(weights=[1, 0.5, 0, 0.25, 0.1], coefficient_type="chebyshev", tol=None, max_iters=12, error_type="iters")),
    ("rmat10/pagerank_closed", "rmat10_dir", "pagerank_closed", dict(alpha=0.85, tol=1e-9, max_iters=1000)),
    ("rmat10/absorbing_085", "rmat10_dir", "absorbing", dict(alpha=0.85, max_iters=1000)),
    ("rmat10/absorbing_custom", "rmat10_dir", "absorbing", dict(alpha=0.9, tol=1e-9, max_iters=1000, _absorption="ramp")),
    ("rmat10/lowpass", "rmat10_dir", "lowpass", dict(params=[0.9] * 10)),
    # ---- undirected RMAT (symmetric normalisation by "auto") ----
    ("rmat12/pagerank_default", "rmat12_sym", "pagerank", dict(alpha=0.85)),
    ("rmat12/pagerank_tol1e-9", "rmat12_sym", "pagerank", dict(alpha=0.85, tol=1e-9, max_iters=1000)),
    ("rmat12/heat_taylor", "rmat12_sym", "heat", dict(**_HK)),
    ("rmat12/heat_cheb", "rmat12_sym", "heat", dict(coefficient_type="chebyshev", **_HK)),
    ("rmat12/absorbing_085", "rmat12_sym", "absorbing", dict(alpha=0.85, tol=1e-9, max_iters=1000)),
    ("rmat12/pagerank_eigenvectors", "rmat12_sym", "pagerank", dict(alpha=0.85, converge_to_eigenvectors=True, tol=1e-9, max_iters=2000)),
    # ---- SymmetricAbsorbingRandomWalks (adhoc.py:317-369) ----
    ("er10k/sarw_default", "er10k", "sarw", dict(max_iters=1000)),
    ("rmat10/sarw_l1", "rmat10_dir", "sarw", dict(error_type="l1", tol=1e-7, max_iters=1000)),
    ("rmat12/sarw_tol1e-9", "rmat12_sym", "sarw", dict(tol=1e-9, max_iters=1000)),
    ("w300/sarw_noquot", "weighted300", "sarw", dict(use_quotient=False, tol=1e-10, max_iters=1000)),
    # ---- weighted digraph, non-uniform personalization (norm preserved) ----
    ("w300/pagerank_default", "weighted300", "pagerank", dict(alpha=0.85)),
    ("w300/pagerank_tol1e-10", "weighted300", "pagerank", dict(alpha=0.85, tol=1e-10, max_iters=1000)),
    ("w300/pagerank_both", "weighted300", "pagerank", dict(alpha=0.85, normalization="both", tol=1e-9, max_iters=1000)),
    ("w300/heat_taylor", "weighted300", "heat", dict(**_HK)),
    ("w300/absorbing_085", "weighted300", "absorbing", dict(alpha=0.85, tol=1e-10, max_iters=1000)),
    ("w300/pagerank_nonorm", "weighted300", "pagerank", dict(alpha=0.85, preserve_norm=False, tol=1e-10, max_iters=1000)),
]

NORMALIZATIONS = ["col", "symmetric", "both", "laplacian", "none"]


def absorption_vector(kind, n):
    if kind == "ramp":
        return 0.5 + np.arange(n, dtype=np.float64) / n
    raise KeyError(kind)
