"""CPU validation of the merge-path SpMV algorithm (tests/merge_path_model.py mirrors the HIP kernels)."""
import numpy as np
import pytest
import scipy.sparse as sp

from merge_path_model import spmv_model


def _random_csr(rng, n, density, hub_rows=0, empty_frac=0.0):
    A = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="lil")
    for h in range(hub_rows):
        r = int(rng.integers(0, n))
        A[r, :] = rng.random(n)
    A = A.tocsr()
    if empty_frac > 0:
        kill = rng.random(n) < empty_frac
        A = sp.diags((~kill).astype(float)) @ A
        A = sp.csr_array(A)
        A.eliminate_zeros()
    A = sp.csr_array(A)
    A.sort_indices()
    return A


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("ipt,wg,wave", [(3, 16, 4), (1, 8, 4), (7, 8, 2), (2, 4, 4)])
def test_model_matches_scipy(seed, ipt, wg, wave):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 120))
    A = _random_csr(rng, n, density=float(rng.choice([0.0, 0.01, 0.05, 0.3])), hub_rows=int(rng.integers(0, 3)),
                    empty_frac=float(rng.choice([0.0, 0.5, 0.9])))
    x = rng.random(n)
    y = spmv_model(A.indptr, A.indices, A.data, x, ipt=ipt, wg=wg, wave=wave)
    assert np.allclose(y, A @ x, rtol=1e-12, atol=1e-14)


def test_model_degenerate_shapes():
    for n in (1, 2, 5):
        A = sp.csr_array((n, n))
        assert np.array_equal(spmv_model(A.indptr, A.indices, A.data, np.ones(n)), np.zeros(n))
    # one dense row spanning many tiles, surrounded by empty rows
    n = 40
    A = sp.lil_array((n, n))
    A[17, :] = np.arange(1, n + 1)
    A = sp.csr_array(A.tocsr())
    x = np.linspace(1, 2, n)
    assert np.allclose(spmv_model(A.indptr, A.indices, A.data, x, ipt=2, wg=4, wave=2), A @ x)
    # every row dense
    A = sp.csr_array(np.arange(1.0, 1 + 30 * 30).reshape(30, 30))
    x = np.linspace(-1, 1, 30)
    assert np.allclose(spmv_model(A.indptr, A.indices, A.data, x, ipt=3, wg=8, wave=4), A @ x)
