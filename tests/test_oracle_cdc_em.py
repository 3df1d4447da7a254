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


@pytest.mark.parametrize("dof", [12, 6])
def test_initialize_against_numpy(ob, dof):
    """without normalisation the fit is a few lines of numpy: nearest neighbours, normals from numpy's eigh, 2 x MUSE, Beaton-Tukey,
    the inverse of sum J^T n n^T J w"""
    mv, fx, T0, Ct = scenario(dof, n=3000)
    c = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=dof)
    scale = c.initialize(normalize=False)
    y = (mv.loc - T0.c) @ T0.A.T + T0.t
    to = ob.nn(fx.loc, y, ob.NN_BRUTE)
    lam, vec = np.linalg.eigh(fx.cov.reshape(-1, 3, 3))
    nrm = vec[:, :, 0][to]                       # eigenvalues ascending: column 0 is the normal
    e = y - fx.loc[to]
    r = np.abs(np.einsum("ij,ij->i", e, nrm))
    assert scale == pytest.approx(2.0 * ob.muse_scale(r), rel=1e-12)
    u = (r / scale) ** 2
    w = np.array([ob.weight_BT(v) for v in u]) / scale ** 2
    assert 0 < np.count_nonzero(w) <= mv.n
    d = mv.loc - T0.c
    J = np.zeros((mv.n, 3, dof))
    if dof == 12:   # (cdcl_trans_affine.txx:141-161) rows of A, then t
        for k in range(3):
            J[:, k, 3 * k:3 * k + 3] = d
            J[:, k, 9 + k] = 1.0
    else:           # (cdcl_trans_rigid3d.cxx:197-225) d(R(x - c))/d(alpha, beta, gamma) at the current angles is not needed for a
        # numpy check of this size: compare the translation block only
        J = None
    if J is not None:
        nJ = np.einsum("ik,ikd->id", nrm, J)
        AtA = np.einsum("i,id,ie->de", w, nJ, nJ)
        assert np.allclose(c.cov, np.linalg.pinv(AtA, rcond=1e-15, hermitian=True), rtol=1e-6, atol=1e-12 * np.abs(c.cov).max())
    else:
        # translation block of the information matrix: sum w n n^T (the Jacobian of the translation is the identity)
        info = np.linalg.pinv(c.cov, rcond=1e-15, hermitian=True)
        assert np.allclose(info[3:, 3:], np.einsum("i,ij,ik->jk", w, nrm, nrm), rtol=1e-6)
    assert np.allclose(c.cov, c.cov.T, atol=1e-12 * np.abs(c.cov).max())


@pytest.mark.parametrize("dof", [12, 6])
def test_initialize_normalised_is_the_same_fit_in_other_units(ob, dof):
    """normalize_matches() changes the coordinates the fit is made in (and with them the MUSE scale in mm), not the pairs"""
    from lrr_amd import synth
    mv, fx, T0, Ct = scenario(dof, n=3000)
    if dof == 6:
        # cdcl_trans_rigid3d::normalize keeps center_moving_ in mm beside normalised features (cdcl_trans_rigid3d.cxx:260-280): the
        # normalised rigid map is only meaningful with the centre at the origin, so the same map is written that way here
        T0 = synth.Transform(T0.A, T0.t - T0.A @ T0.c, np.zeros(3))
    a = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=dof)
    b = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=dof)
    sa, sb = a.initialize(normalize=False), b.initialize(normalize=True)
    assert sa > 0 and sb > 0 and sb < sa      # radii of tens of mm: the normalised residuals are that much smaller
    assert np.all(np.isfinite(b.cov)) and np.all(np.linalg.eigvalsh(0.5 * (b.cov + b.cov.T)) > -1e-12)
    # without normalisation the transform is not touched; with it, normalize() / unnormalize() hand back the same map about the
    # moving centroid (cdcl_trans_affine.txx:259-286, cdcl_trans_rigid3d.cxx:284-305)
    assert np.array_equal(a.A, T0.A) and np.array_equal(a.t, T0.t) and np.array_equal(a.c, T0.c)
    assert np.allclose(b.c, mv.loc.mean(axis=0), atol=1e-9) and np.allclose(b.A, T0.A, atol=1e-12)
    x = mv.loc[:50]
    if dof == 12:
        assert np.allclose((x - b.c) @ b.A.T + b.t, (x - T0.c) @ T0.A.T + T0.t, atol=1e-9)
    else:
        # cdcl_trans_rigid3d::unnormalize computes the translation of a map about the origin and then sets the centre to the moving
        # centroid (cdcl_trans_rigid3d.cxx:294-304): after a normalised initialize() the rigid transform is another map.  Restated
        # as it is; the ICP classes only ever normalise the increment, which they recompose without its centre.
        assert np.allclose(b.t, T0.t, atol=1e-9)


def test_run_is_initialize_plus_iterations(ob):
    mv, fx, T0, Ct = scenario(12, n=1500)
    a = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=12)
    conv, its = a.run(max_iterations=8)
    b = ob.CdcEmOracle(mv, fx, T0.A, T0.t, T0.c, Ct, seed=77, dof=12)
    b.initialize()
    for it in range(its):
        b.one_iteration(it)
    assert its <= 8 and np.array_equal(a.A, b.A) and np.array_equal(a.cov, b.cov)
