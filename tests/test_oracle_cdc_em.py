"""The oracle's restatement of cdcl_estimation::one_iteration (oracle/lrr_oracle.c: lrr_oracle_cdc_em_iteration_pyramid,
after cdcl/cdcl_estimation.txx:95-187, 342-545) on its own, on the CPU: the control flow the GPU parity test
(tests/test_gpu_cdc.py::test_em_iterations_equal_the_oracle) relies on really happens in the scenarios it uses — the
resolution switches in both directions, the closest-point fallback, a rigid parameter estimate that moves.  The
reference holds no fixture for this path (its tests need the third-party optimiser and the data files of
cdcl/exe): parity unpinned beyond the structure checked here."""
import numpy as np
import pytest

from tests.cdc_scenarios import EM_CASES, scenario


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.build()
    return binding


@pytest.mark.parametrize("name,dof,kw,spacing,its,switches", [c for c in EM_CASES if c[3] is not None], ids=[c[0] for c in EM_CASES if c[3] is not None])
def test_resolution_switches_both_ways(ob, name, dof, kw, spacing, its, switches):
    mvs, fxs, T0, Ct = scenario(dof, **kw)
    c = ob.CdcEmOracle(mvs, fxs, T0.A, T0.t, T0.c, Ct, seed=77, dof=dof, spacing=spacing)
    assert c.resolution == 1          # the coarsest first (cdcl/cdcl_estimation_ICP.txx:118)
    down = up = 0
    for it in its:
        c.one_iteration(it)
        if it <= 5:
            assert c.switches == (0, 0) and c.resolution == 1   # (resolution_switch_allowed = iteration > startEM, :105)
        # one restart per compute_weights, two compute_weights per iteration
        assert c.switches[0] + c.switches[1] <= 2
        down, up = down + c.switches[0], up + c.switches[1]
        assert c.rc.shape[0] == mvs[c.resolution].n and c.matches_size == int((c.rc > 0).sum())
    assert (down, up) == switches


@pytest.mark.parametrize("dof", [12, 6])
def test_closest_point_fallback(ob, dof):
    """(:543-544, cdcl/cdcl_estimation_ICP.txx:451-489) no match inside any search radius: every moving feature is paired with
    its nearest fixed feature under trans_, weight 1"""
    mv, fx, T0, Ct = scenario(dof, n=1500, cov_scale=2e-4, shift=np.array([150.0, 0.0, 0.0]))
    c = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=dof)
    c.one_iteration(0)
    assert c.fell_back and c.matches_size == mv.n and np.all(c.rc == 1) and np.all(c.w[:, 0] == 1.0)
    assert np.isinf(c.avg_largest_eval)       # (:491: a sum over all moving features divided by zero matches)


def test_rigid_parameter_estimate_moves(ob):
    mv, fx, T0, Ct = scenario(6, n=3000)
    c = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=6)
    for it in range(6):
        c.one_iteration(it)
    A0, t0 = c.A.copy(), c.t.copy()
    conv, info = c.one_iteration(6)
    assert info[6] >= 2 and info[1] < info[0] and info[4] > 0
    assert not np.allclose(A0, c.A, atol=1e-9) or not np.allclose(t0, c.t, atol=1e-9)
    # the global transform stays a rotation (recompose_increment orthogonalises the increment, cdcl_trans_rigid3d.cxx:309-325)
    assert np.allclose(c.A @ c.A.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(c.A) - 1.0) < 1e-12
