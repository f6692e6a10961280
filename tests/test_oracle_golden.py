"""Pins the oracle (oracle/ref_loops.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

import cases
from oracle import ref_loops as orc

TOL = 1e-12   # fp64 restatement vs fp64 reference: relative L-inf


def run_oracle(A, directed, p, algo, kwargs):
    kwargs = dict(kwargs)
    absorption = kwargs.pop("_absorption", None)
    pre = {k: kwargs.pop(k) for k in ("normalization", "renormalize") if k in kwargs}
    M = orc.normalize(A, pre.get("normalization", "auto"), directed, pre.get("renormalize", 0.0))
    if algo == "pagerank":
        return orc.pagerank(M, p, **kwargs)
    if algo == "heat":
        t = kwargs.pop("t", 3)
        return orc.heat_kernel(M, p, t=t, **kwargs)
    if algo == "generic":
        return orc.generic_filter(M, p, kwargs.pop("weights"), **kwargs)
    if algo == "pagerank_closed":
        return orc.pagerank_closed(M, p, kwargs.pop("alpha"), **kwargs)
    if algo == "absorbing":
        if absorption is not None:
            kwargs["absorption"] = cases.absorption_vector(absorption, A.shape[0])
        return orc.absorbing_walks(M, p, **kwargs)
    if algo == "lowpass":
        return orc.low_pass_recursive(M, p, kwargs.pop("params"), **kwargs)
    if algo == "sarw":
        return orc.symmetric_absorbing_walks(M, p, **kwargs)
    raise KeyError(algo)


@pytest.mark.parametrize("name,gkey,algo,kwargs", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_oracle_matches_reference(golden, graphs, name, gkey, algo, kwargs):
    A, directed, p = graphs(gkey)
    ranks, iters = run_oracle(A, directed, p, algo, kwargs)
    want = golden[name + "|ranks"]
    assert iters == int(golden[name + "|iters"])
    assert np.max(np.abs(ranks - want)) <= TOL * np.max(np.abs(want))


def test_cfg1_scalars_from_survey(golden):
    """SURVEY.md 8c probe values for BASELINE.json configs[0]."""
    r = golden["er10k/pagerank_default|ranks"]
    assert int(golden["er10k/pagerank_default|iters"]) == 10
    assert abs(r.sum() - 3.0) < 1e-12
    assert abs(r.max() - 0.16204511522684384) < 1e-15
    assert np.allclose(r[:3], [0.16204512, 0.16037664, 0.16060059], atol=1e-8)
    assert int(golden["er10k/pagerank_tol1e-9|iters"]) == 18
    assert int(golden["er10k/pagerank_noquot|iters"]) == 25
    assert abs(golden["er10k/pagerank_noquot|ranks"].sum() - 2.7448368603781828) < 1e-12
    assert abs(golden["er10k/heat_taylor|ranks"].sum() - 31.097145532821934) < 1e-10
    assert abs(golden["er10k/heat_cheb|ranks"].sum() - 30.988269660250896) < 1e-10
    assert int(golden["er10k/absorbing_default|iters"]) == 21


def test_max_iters_raises_and_zero_input(golden, graphs):
    assert int(golden["rmat10/max_iters_raises"]) == 1
    assert float(golden["rmat10/zero_personalization_sum"]) == 0.0
    A, directed, p = graphs("rmat10_dir")
    M = orc.normalize(A, "auto", directed)
    with pytest.raises(Exception):
        orc.pagerank(M, p, max_iters=5, tol=1e-12)
    r, it = orc.pagerank(M, np.zeros(len(p)))
    assert it == 0 and r.sum() == 0


def test_residuals(golden):
    u, v = golden["residual|u"], golden["residual|v"]
    assert orc.mabs(u, v) == float(golden["residual|mabs"])
    assert orc.l1(u, v) == float(golden["residual|l1"])
    assert orc.maxdiff(u, v) == float(golden["residual|linf"])


@pytest.mark.parametrize("gkey", ["rmat10_dir", "weighted300"])
@pytest.mark.parametrize("normalization", cases.NORMALIZATIONS)
@pytest.mark.parametrize("renorm", [0, 1])
def test_normalisation(golden_norm, graphs, gkey, normalization, renorm):
    A, directed, _ = graphs(gkey)
    M = orc.normalize(A, normalization, directed, renorm).tocsr()
    M.sort_indices()
    key = f"{gkey}|{normalization}|{renorm}"
    assert np.array_equal(M.indptr, golden_norm[key + "|indptr"])
    assert np.array_equal(M.indices, golden_norm[key + "|indices"])
    assert np.allclose(M.data, golden_norm[key + "|data"], rtol=1e-15, atol=0)
    assert np.allclose(orc.row_sums(M), golden_norm[key + "|degrees"], rtol=1e-14, atol=1e-300)
    x = np.linspace(0.1, 1.0, A.shape[0])
    assert np.allclose(orc.conv(x, M), golden_norm[key + "|conv"], rtol=1e-13, atol=1e-300)


def test_rmat_generator_is_deterministic():
    from oracle import rmat_np
    s1, d1 = rmat_np.rmat_edges(8, 4, seed=5)
    s2, d2 = rmat_np.rmat_edges(8, 4, seed=5, first_edge=100, num_edges=50)
    assert np.array_equal(s1[100:150], s2) and np.array_equal(d1[100:150], d2)
    assert s1.max() < 256 and d1.max() < 256
    # quadrant frequencies of the top bit follow (a, b, c, d)
    s, d = rmat_np.rmat_edges(12, 16, seed=0)
    top = ((s >> 11) << 1) | (d >> 11)
    freq = np.bincount(top, minlength=4) / len(top)
    assert np.allclose(freq, [0.57, 0.19, 0.19, 0.05], atol=0.01)
