"""SURVEY.md section 8(f)4, first step: the E-step of the covariance-driven EM estimator,
cdcl_estimation::compute_weights (cdcl/cdcl_estimation.txx:342-545) — radius search under the transfer-error
covariance, at most 100 candidates per moving feature, Mahalanobis weights, two-sided normalisation — on the device
(lrr_icp_cdc_compute_weights) against the CPU oracle's restatement.  Match sets exact, weights to 1e-8.  The third-party
closed-form eigenvalue routine behind the search radius (vnl) is restated, not verified: parity unpinned."""
import dataclasses

import numpy as np
import pytest

from lrr_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from lrr_amd import estimation
    estimation.hip_lib()
    estimation.set_device(0)
    return estimation


@pytest.fixture(scope="module")
def ob():
    from oracle import binding
    binding.build()
    return binding


def _rows(rc, ti, w):
    out = []
    for i in range(rc.shape[0]):
        k = int(rc[i])
        o = np.argsort(ti[i, :k], kind="stable")
        out.append((ti[i, :k][o], w[i, :k][o]))
    return out


@pytest.mark.parametrize("dof,n,param_sigma,max_matches", [(12, 4000, 2e-3, 100), (6, 3000, 1e-3, 100), (12, 6000, 4e-3, 20)])
def test_compute_weights_equals_oracle(E, ob, dof, n, param_sigma, max_matches):
    pair = synth.make_pair(n, n - 500, 717, kind="rigid" if dof == 6 else "affine", extent=60.0)
    mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
    fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
    T0 = synth.perturbed(pair.planted, 5, rot_deg=0.3, trans_mm=0.4)
    rng = np.random.default_rng(2)
    B = rng.normal(size=(dof, dof)) * param_sigma
    if dof == 12:
        B[9:] *= 50.0   # translations are far less certain than the matrix entries
    else:
        B[3:] *= 50.0
    Ct = B @ B.T        # parameter covariance: the search radius grows with the distance from the centre
    want = ob.cdc_compute_weights(mv, fx, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=max_matches)
    cm, cf = E.DeviceCloud(mv), E.DeviceCloud(fx)
    cm.set_covariances(mv.cov)
    cf.set_covariances(fx.cov)
    rc, ti, w, over = E.cdc_compute_weights(cm, cf, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=max_matches)
    assert over == 0
    assert np.array_equal(rc, want[0])
    assert rc.max() <= max_matches and rc.max() >= min(max_matches, 10) and (rc > 0).mean() > 0.5
    if max_matches == 20:
        assert (rc == 20).any()  # the selection among more candidates than allowed was exercised
    got_rows, want_rows = _rows(rc, ti, w), _rows(*want)
    for i, ((gi, gw), (wi, ww)) in enumerate(zip(got_rows, want_rows)):
        assert np.array_equal(gi, wi), i
        assert np.allclose(gw, ww, rtol=1e-8, atol=1e-300), i  # column sums are 2^-50 fixed point on the device
    # two-sided normalisation: every weight is w0^2 / (row sum * column sum) -> all in (0, 1]
    allw = np.concatenate([r[1] for r in got_rows])
    assert allw.min() > 0.0 and allw.max() <= 1.0 + 1e-9
    # deterministic (integer atomics for the column sums, fixed-order row sums)
    rc2, ti2, w2, _ = E.cdc_compute_weights(cm, cf, dof, T0.A, T0.t, T0.c, Ct, seed=33, max_matches=max_matches)
    for (gi, gw), (hi, hw) in zip(got_rows, _rows(rc2, ti2, w2)):
        assert np.array_equal(gi, hi) and np.array_equal(gw, hw)
    cm.close()
    cf.close()


def test_compute_weights_needs_covariances(E):
    pair = synth.make_pair(200, 200, 5)
    a, b = E.DeviceCloud(pair.moving), E.DeviceCloud(pair.fixed)
    with pytest.raises(RuntimeError, match="covariances"):
        E.cdc_compute_weights(a, b, 12, np.eye(3), np.zeros(3), np.zeros(3), np.eye(12) * 1e-6, seed=1)
    a.close()
    b.close()
