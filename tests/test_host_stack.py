"""CPU tests of the host-side Python (backend module, signals, preprocessor, convergence, filters) driven
through the C-ABI against the host test double (oracle/host_abi.cpp).  The same tests run on the real
engine in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import cases
from parity_common import EPS32, rel_linf, run_engine, run_oracle, tol_is_fp32_safe, tolerance_for


@pytest.mark.parametrize("name,gkey,algo,kwargs", cases.CASES, ids=[c[0] for c in cases.CASES])
def test_fused_route_matches_reference(host_engine, golden, graphs, name, gkey, algo, kwargs):
    A, directed, p = graphs(gkey)
    got, iters, ranker = run_engine(host_engine, A, directed, p, algo, kwargs)
    # expectation at the engine's effective tolerance max(tol, eps_fp32)  (convergence.py:101)
    want, want_iters = run_oracle(A, directed, p, algo, kwargs, eps=EPS32)
    assert iters == want_iters
    assert rel_linf(got, want) <= tolerance_for(kwargs)
    if tol_is_fp32_safe(kwargs):          # then the committed golden vector of the reference applies as-is
        assert iters == int(golden[name + "|iters"])
        assert rel_linf(got, golden[name + "|ranks"]) <= tolerance_for(kwargs)
    if algo != "lowpass":
        assert hasattr(ranker, "last_loop"), "fused device loop was expected to run"
