"""The oracle against the committed golden fixtures (tests/golden/make_golden.py).  The fixtures
were produced with the brute-force NN back-end; here the kd-tree back-end replays them, so this
also pins kd-tree == brute force inside full estimator runs."""
import pytest

from oracle import binding as ob
from tests.common import GOLDEN_CASES, check_against_golden, load_golden, make_estimator


def test_golden_fixtures_present():
    assert len(GOLDEN_CASES) >= 7


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_oracle_reproduces_golden(oracle_lib, name):
    g, mv, fx = load_golden(name)
    e = make_estimator(ob.OracleEstimator, g, mv, fx, nn_backend=ob.NN_KDTREE)
    check_against_golden(e, g, match_exact=True, rtol=1e-12, param_rtol=1e-12)
