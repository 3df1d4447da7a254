"""Shared helpers: golden fixtures and a backend-agnostic replay of an estimator run."""
import glob
import os

import numpy as np

from lrr_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                      if not os.path.basename(p).startswith("features_"))  # estimator runs (the features_* files belong to the .vtp fixtures)


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    mv = synth.FeatureSet(g["moving_loc"], g["moving_proj"], g["moving_strength"], g["moving_shape"], g["moving_dirs"])
    fx = synth.FeatureSet(g["fixed_loc"], g["fixed_proj"], g["fixed_strength"], g["fixed_shape"], g["fixed_dirs"])
    return g, mv, fx


def make_estimator(factory, g, mv, fx, **kw):
    """factory: OracleEstimator or lrr_amd.estimation.Estimator (same constructor and methods)."""
    kind, dof, seed = int(g["kind"]), int(g["dof"]), int(g["seed"])
    sym = kind >= 2
    e = factory(kind, dof, mv, fx, g["A0"], g["t0"], g["c0"], moving_inside=mv if sym else None,
                fixed_inside=fx if sym else None, seed=seed, **kw)
    if not sym:
        if bool(g["normalize"]):
            e.normalize_matches()
        e.weight_by_strength(bool(g["strength"]))
    inject = float(g["inject"])
    if inject > 0:
        e.inject_initial_scale(inject, inject * 1.1)
    e.set_transform(g["A0"], g["t0"], g["c0"])
    if not sym:
        e.set_number_points(mv.n)
    return e


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check_against_golden(e, g, match_exact=True, rtol=1e-9, param_rtol=1e-7):
    """Replays the golden run on estimator e and compares every recorded stage."""
    sym = int(g["kind"]) >= 2
    for it in range(int(g["iterations"])):
        conv = e.one_iteration(it)
        assert conv == bool(g[f"it{it}_converged"]), f"iteration {it}: converged flag"
        assert e.number_points(0) == int(g[f"it{it}_number_points"]), f"iteration {it}: schedule"
        for d in ((0, 1) if sym else (0,)):
            f, t, w = e.matches(d)
            if match_exact:
                assert np.array_equal(f, g[f"it{it}_d{d}_from"]), f"it {it} dir {d}: from indices"
                assert np.array_equal(t, g[f"it{it}_d{d}_to"]), f"it {it} dir {d}: match set"
            gw = g[f"it{it}_d{d}_w"]
            assert np.allclose(w, gw, rtol=rtol, atol=1e-12 * np.abs(gw).max()), f"it {it} dir {d}: weights"
            A, tt, c, cov = e.get_transform(d)
            assert rel(A, g[f"it{it}_d{d}_A"]) < param_rtol, f"it {it} dir {d}: A"
            assert rel(tt, g[f"it{it}_d{d}_t"]) < param_rtol, f"it {it} dir {d}: t"
            assert np.allclose(c, g[f"it{it}_d{d}_c"], rtol=1e-9, atol=1e-9), f"it {it} dir {d}: centre"
            assert rel(cov, g[f"it{it}_d{d}_cov"]) < 1e-5, f"it {it} dir {d}: covariance"
            assert e.last_scale(d) == np.float64(g[f"it{it}_d{d}_scale"]) or \
                abs(e.last_scale(d) - float(g[f"it{it}_d{d}_scale"])) <= rtol * abs(float(g[f"it{it}_d{d}_scale"])), f"it {it} dir {d}: scale"
            assert abs(e.weighted_error(d) - float(g[f"it{it}_d{d}_werr"])) <= 1e-8 * max(1.0, abs(float(g[f"it{it}_d{d}_werr"]))), f"it {it} dir {d}: weighted error"
        AtA, Atb = e.normal_equations()
        assert rel(AtA, g[f"it{it}_AtA"]) < rtol, f"it {it}: AtA"
        assert rel(Atb, g[f"it{it}_Atb"]) < rtol, f"it {it}: Atb"
        assert abs(e.RMS_error() - float(g[f"it{it}_rms"])) <= 1e-8 * max(1.0, abs(float(g[f"it{it}_rms"]))), f"it {it}: RMS"
