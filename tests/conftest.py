import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def _oracle_module():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture
def oracle(_oracle_module):
    # the per-class rho multipliers and the step-size schedule are process-global in the oracle:
    # every test starts from OSQP's uniform rho without a schedule and opts in to an engine's step
    # sizes with match_engine()
    _oracle_module.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)
    _oracle_module.set_rho_phases(None)
    yield _oracle_module
    _oracle_module.set_rho_class_scale(1.0, 1.0, 1.0, 1.0)
    _oracle_module.set_rho_phases(None)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
