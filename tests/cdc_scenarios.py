"""Inputs of the CDC-EM tests, shared by the CPU tests of the oracle and the GPU parity tests:
a planted pair with feature covariances, a perturbed start, a parameter covariance, optionally a two-resolution pyramid
([0] the finest) whose fine moving set carries stray features no fixed feature answers — they raise avg_largest_eval at
the fine level (cdcl/cdcl_estimation.txx:384, :491 divide the sum over ALL moving features by the matches found), which is
what makes the estimator switch back to the coarser level (:512-516)."""
import dataclasses

import numpy as np

from lrr_amd import synth


def with_strays(fs, k, offset):
    extra = fs.subset(np.arange(k))
    extra = dataclasses.replace(extra, loc=extra.loc + offset)
    out = {}
    for f in dataclasses.fields(fs):
        a, b = getattr(fs, f.name), getattr(extra, f.name)
        out[f.name] = np.concatenate([a, b]) if isinstance(a, np.ndarray) else a
    return type(fs)(**out)


def scenario(dof, n, cov_seed=4, cov_scale=2e-3, shift=0.0, pyramid=False, strays=0):
    pair = synth.make_pair(n, n - 300, 313, kind="affine" if dof == 12 else "rigid", extent=60.0)
    mv = dataclasses.replace(pair.moving, cov=synth.covariances(pair.moving))
    fx = dataclasses.replace(pair.fixed, cov=synth.covariances(pair.fixed))
    T0 = synth.perturbed(pair.planted, 5, rot_deg=0.4, trans_mm=0.5)
    T0 = synth.Transform(T0.A, T0.t + shift, T0.c)
    B = np.random.default_rng(cov_seed).normal(size=(dof, dof)) * cov_scale
    B[dof - 3:] *= 50.0
    Ct = B @ B.T
    if not pyramid:
        return mv, fx, T0, Ct
    rng = np.random.default_rng(9)
    mvs = [mv, mv.subset(np.sort(rng.choice(mv.n, mv.n // 4, replace=False)))]
    fxs = [fx, fx.subset(np.sort(rng.choice(fx.n, fx.n // 4, replace=False)))]
    if strays:
        mvs[0] = with_strays(mvs[0], strays, 150.0)
    return mvs, fxs, T0, Ct


# (id, dof, scenario arguments, fixed spacings, iterations to run, resolution switches expected over the run: to finer, to coarser)
EM_CASES = [
    ("affine", 12, dict(n=5000), None, tuple(range(9)), (0, 0)),
    ("rigid", 6, dict(n=3000), None, tuple(range(9)), (0, 0)),
    ("affine-pyramid", 12, dict(n=3000, cov_seed=6, pyramid=True, strays=600), [0.15, 0.3], (0, 1, 6, 7, 8), (3, 2)),
    ("rigid-pyramid", 6, dict(n=3000, cov_seed=4, pyramid=True, strays=600), [0.3, 0.6], (0, 1, 6, 7, 8), (3, 3)),
]
