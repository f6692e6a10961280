"""Backend-module / signal / preprocessor behaviour, restating the assertions of the reference's
tests/test_core.py:25-155, tests/test_preprocessor.py:6-46 and the identities of tests/test_filters.py on
synthetic graphs (the reference's datasets are downloads that do not exist offline).
Runs on the host test double here; tests/test_gpu_parity.py re-runs the same bodies on the MI355X."""
import numpy as np
import pytest

import core_checks


@pytest.mark.parametrize("check", core_checks.ALL, ids=[c.__name__ for c in core_checks.ALL])
def test_core(host_engine, check):
    check(host_engine)
