"""Runs the kernel-level checks against the host test double (validates the checks and the double on CPU)."""
import pytest

import kernel_checks


@pytest.mark.parametrize("check", kernel_checks.ALL, ids=[c.__name__ for c in kernel_checks.ALL])
def test_kernel_checks_on_double(host_engine, check):
    check(host_engine)
