"""Known-answer tests that pin the CPU oracle's building blocks (the reference has no
tests or golden vectors for this path: SURVEY.md §4, §8c -> parity unpinned)."""
import numpy as np
import pytest

from oracle import binding as ob


def test_weight_BT_table(oracle_lib):
    # cdcl/cdcl_utils.h:142-152: (1 - u/16)^2 for u <= 16 else 0
    for u, want in [(0.0, 1.0), (4.0, 0.5625), (8.0, 0.25), (16.0, 0.0), (16.0001, 0.0), (100.0, 0.0)]:
        assert ob.weight_BT(u) == pytest.approx(want, abs=1e-15)


def test_mad_scale_formula(oracle_lib):
    rng = np.random.default_rng(0)
    for n in (2, 3, 10, 11, 1000, 1001):
        v = rng.normal(size=n)
        a = np.sort(np.abs(v))
        want = 1.4826 * (1 + 5.0 / (n - 1)) * a[(n - 1) // 2 + 1]
        assert ob.mad_scale(v) == want
    assert ob.mad_scale(np.array([3.0])) == 0.0


def test_muse_recovers_gaussian_sigma(oracle_lib):
    rng = np.random.default_rng(1)
    for sigma in (0.3, 2.0):
        r = np.abs(rng.normal(scale=sigma, size=20000))
        assert ob.muse_scale(r) == pytest.approx(sigma, rel=0.03)
        # 30 % gross outliers: bounded upward bias.  The k-th inlier sits at quantile k/(0.7 n)
        # of the inlier distribution but is priced at k/n, so the smallest fraction (0.25) gives
        # sigma * Phi^-1((1+.25/.7)/2) / Phi^-1((1+.25)/2) = 1.455 sigma at best.
        r2 = r.copy()
        r2[:6000] = rng.uniform(5 * sigma, 50 * sigma, size=6000)
        assert 1.3 * sigma < ob.muse_scale(r2) < 1.7 * sigma
    # scale equivariance
    r = np.abs(rng.normal(size=5000))
    assert ob.muse_scale(3.0 * r) == pytest.approx(3.0 * ob.muse_scale(r), rel=1e-12)


def test_nn_brute_equals_kdtree_with_ties(oracle_lib):
    rng = np.random.default_rng(2)
    f = rng.uniform(0, 50, size=(4000, 3))
    f[100:200] = f[0:100]            # exact duplicates -> distance ties, lowest index must win
    f[300:400] = np.round(f[300:400])  # lattice points -> more ties
    q = np.concatenate([rng.uniform(-10, 60, size=(3000, 3)), f[150:250], np.round(rng.uniform(0, 50, size=(500, 3)))])
    a = ob.nn(f, q, ob.NN_BRUTE)
    b = ob.nn(f, q, ob.NN_KDTREE)
    assert np.array_equal(a, b)
    d = ((f[:, None, :] - q[None, :50, :]) ** 2).sum(-1)
    assert np.array_equal(a[:50], d.argmin(0))
    # a query that IS a duplicated fixed point returns the lower index
    assert np.all(a[3000:3050] < 100)


def test_nn_empty_and_single(oracle_lib):
    f = np.array([[1.0, 2.0, 3.0]])
    q = np.zeros((4, 3))
    assert np.array_equal(ob.nn(f, q, ob.NN_BRUTE), np.zeros(4, dtype=np.int64))
    assert np.array_equal(ob.nn(f, q, ob.NN_KDTREE), np.zeros(4, dtype=np.int64))
    assert ob.nn(f, np.zeros((0, 3))).shape == (0,)
    assert np.array_equal(ob.nn(np.zeros((0, 3)), q, ob.NN_KDTREE), -np.ones(4, dtype=np.int64))


def test_shuffle_is_a_seeded_permutation(oracle_lib):
    a, s1 = ob.shuffle(1000, 42)
    b, s2 = ob.shuffle(1000, 42)
    c, _ = ob.shuffle(1000, 43)
    assert np.array_equal(a, b) and s1 == s2
    assert not np.array_equal(a, c)
    assert np.array_equal(np.sort(a), np.arange(1000))


def test_sym_pinv_and_svd3(oracle_lib):
    rng = np.random.default_rng(3)
    for n in (6, 12):
        B = rng.normal(size=(n + 5, n))
        M = B.T @ B
        assert np.allclose(ob.sym_pinv(M), np.linalg.inv(M), rtol=1e-9, atol=1e-12)
    for _ in range(20):
        M = rng.normal(size=(3, 3))
        U, s, V = ob.svd3(M)
        assert np.allclose(U @ np.diag(s) @ V.T, M, atol=1e-13)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-13)
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose(np.sort(s), np.sort(np.linalg.svd(M, compute_uv=False)), atol=1e-13)


def test_voronoi_volume_lookup_against_numpy(oracle_lib):
    """the oracle's lazily filled 1 mm index volume (compute_distance_map.cxx:55-151 + rrl .txx:530-543) against a
    dense numpy construction of the same volume: every voxel's nearest occupied voxel (exact, ties -> lowest
    feature index), feature 0 never stored, later features overwrite earlier ones, queries clamped to the volume"""
    import numpy as np
    from oracle import binding as ob
    rng = np.random.default_rng(3)
    loc = rng.uniform(10.0, 22.0, size=(300, 3))
    loc[50:80] = loc[20:50] + 0.2  # several features per voxel: the later one stays
    origin = loc.min(0)
    size = np.maximum(np.ceil(loc.max(0) - origin).astype(int), 1)
    vox = np.clip(np.floor((loc - origin) + 0.5).astype(int), 0, size - 1)
    label = -np.ones(size[::-1], dtype=int)  # [z, y, x]
    for i in range(1, loc.shape[0]):
        label[vox[i, 2], vox[i, 1], vox[i, 0]] = i
    sites = np.argwhere(label >= 0)  # (z, y, x)
    site_label = label[sites[:, 0], sites[:, 1], sites[:, 2]]
    order = np.argsort(site_label)
    sites, site_label = sites[order], site_label[order]
    site_xyz = origin + sites[:, ::-1].astype(float)
    q = np.concatenate([rng.uniform(5.0, 27.0, size=(4000, 3)), loc, origin + vox + 0.5])
    qv = np.clip(np.floor((q - origin) + 0.5), 0, size - 1)
    centre = origin + qv
    d2 = ((centre[:, None, :] - site_xyz[None, :, :]) ** 2).sum(-1)
    want = site_label[d2.argmin(1)]  # argmin returns the first minimum: lowest label among equidistant sites
    got = ob.nn(loc, q, ob.NN_VORONOI)
    assert np.array_equal(got, want)
    assert 0 not in got


@pytest.mark.parametrize("dof", [12, 6])
def test_cdc_objective_gradient_is_the_derivative_of_the_value(dof):
    """cdcl_obj_fun_par::compute (cdcl/cdcl_obj_fun_par.txx:44-142) as the oracle restates it: the analytic gradient
    (2 J_th^T C_ij^-1 e and the two correction terms of d C_ij / d theta) equals central differences of the value in
    the transform's parameters (affine: A row-major, t; rigid: the small-angle vector of R = I + [w]x, t)."""
    import dataclasses
    from lrr_amd import synth
    n = 700
    pair = synth.make_pair(n, n - 100, 717, kind="rigid" if dof == 6 else "affine", extent=40.0)
    mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
    fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
    T0 = synth.perturbed(pair.planted, 5, rot_deg=0.3, trans_mm=0.4)
    rng = np.random.default_rng(2)
    B = rng.normal(size=(dof, dof)) * 2e-3
    B[(9 if dof == 12 else 3):] *= 50.0
    Ct = B @ B.T
    rc, ti, w = ob.cdc_compute_weights(mv, fx, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=100)
    assert rc.sum() > 1000

    def skew(v):
        return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])

    if dof == 12:
        x0 = np.r_[np.asarray(T0.A, float).ravel(), T0.t]
        unpack = lambda x: (x[:9].reshape(3, 3), x[9:])
    else:
        x0 = np.r_[0.004, -0.003, 0.002, T0.t]   # set_parameterization: R = I + [w]x (cdcl_trans_rigid3d.cxx:140-166)
        unpack = lambda x: (np.eye(3) + skew(x[:3]), x[3:])
    A0, t0 = unpack(x0)
    f, g, rows = ob.cdc_obj_par(mv, fx, dof, A0, t0, T0.c, Ct, rc, ti, w)
    assert rows == int((rc > 0).sum()) and np.isfinite(f)
    num = np.zeros(dof)
    for k in range(dof):
        h = 1e-6
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h
        xm[k] -= h
        fp = ob.cdc_obj_par(mv, fx, dof, *unpack(xp), T0.c, Ct, rc, ti, w, want_grad=False)[0]
        fm = ob.cdc_obj_par(mv, fx, dof, *unpack(xm), T0.c, Ct, rc, ti, w, want_grad=False)[0]
        num[k] = (fp - fm) / (2 * h)
    assert np.abs(num - g).max() <= 1e-6 * np.abs(g).max()


@pytest.mark.parametrize("dof", [12, 6])
def test_cdc_covariance_objective_gradient_is_the_derivative_of_the_value(dof):
    """cdcl_obj_fun_cov::compute (cdcl/cdcl_obj_fun_cov.txx:42-134) as the oracle restates it: the gradient in the entries
    of the Cholesky factor of the parameter covariance equals central differences of the value."""
    import dataclasses
    from lrr_amd import synth
    n = 600
    pair = synth.make_pair(n, n - 100, 717, kind="rigid" if dof == 6 else "affine", extent=40.0)
    mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
    fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
    T0 = synth.perturbed(pair.planted, 5, rot_deg=0.3, trans_mm=0.4)
    rng = np.random.default_rng(2)
    B = rng.normal(size=(dof, dof)) * 2e-3
    B[(9 if dof == 12 else 3):] *= 50.0
    Ct = B @ B.T
    rc, ti, w = ob.cdc_compute_weights(mv, fx, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=100)
    x0 = np.linalg.cholesky(Ct).T[np.triu_indices(dof)]
    f, g, pairs = ob.cdc_obj_cov(mv, fx, dof, T0.A, T0.t, T0.c, x0, rc, ti, w)
    assert pairs == int(rc.sum()) and np.isfinite(f)
    h = 1e-6 * np.abs(x0).max()
    num = np.zeros_like(x0)
    for k in range(len(x0)):
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h
        xm[k] -= h
        num[k] = (ob.cdc_obj_cov(mv, fx, dof, T0.A, T0.t, T0.c, xp, rc, ti, w, want_grad=False)[0]
                  - ob.cdc_obj_cov(mv, fx, dof, T0.A, T0.t, T0.c, xm, rc, ti, w, want_grad=False)[0]) / (2 * h)
    assert np.abs(num - g).max() <= 1e-6 * np.abs(g).max()
