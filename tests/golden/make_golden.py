"""Generates tests/golden/*.npz from the CPU oracle.

The reference holds no golden vectors for this path and cannot be built or
imported here (C++ on VXL/ITK), so these fixtures are SELF-CONSISTENT oracle outputs, not
reference outputs: they pin the oracle against regressions and give the GPU tests committed
expected values.  MUSE-dependent values are marked as such (case names ending in `_muse`);
all other cases inject the initial scale so they do not depend on the unpinned MUSE restatement.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lrr_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (kind, dof, n_fixed, n_moving, seed, planted kind, normalize, strength, inject, iterations)
    "cdcl_forward_rigid": (ob.CDCL_FORWARD, 6, 1500, 1400, 101, "rigid", False, False, 1.5, 4),
    "cdcl_forward_affine_norm_strength": (ob.CDCL_FORWARD, 12, 1500, 1400, 102, "affine", True, True, 0.02, 4),
    "rrl_forward_affine": (ob.RRL_FORWARD, 12, 1500, 1400, 103, "affine", True, True, 0.02, 3),
    "rrl_symmetric_affine": (ob.RRL_SYMMETRIC, 12, 2000, 1900, 104, "affine", True, True, 1.5, 7),
    "cdcl_symmetric_affine": (ob.CDCL_SYMMETRIC, 12, 2000, 1900, 105, "affine", True, True, 1.5, 5),
    "cdcl_symmetric_rigid": (ob.CDCL_SYMMETRIC, 6, 2000, 1900, 106, "rigid", True, True, 1.5, 4),
    "rrl_symmetric_affine_muse": (ob.RRL_SYMMETRIC, 12, 2000, 1900, 107, "affine", True, True, -1.0, 4),
}


def fs_arrays(prefix, fs):
    return {prefix + "_loc": fs.loc, prefix + "_proj": fs.proj, prefix + "_strength": fs.strength,
            prefix + "_shape": fs.shape, prefix + "_dirs": fs.dirs}


def run_case(name):
    kind, dof, nf, nm, seed, pk, normalize, strength, inject, iters = CASES[name]
    pair = synth.make_pair(nf, nm, seed, kind=pk)
    T0 = synth.perturbed(pair.planted, seed)
    sym = kind >= 2
    e = ob.OracleEstimator(kind, dof, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                           moving_inside=pair.moving if sym else None,
                           fixed_inside=pair.fixed if sym else None, seed=seed, nn_backend=ob.NN_BRUTE)
    if not sym:
        if normalize:
            e.normalize_matches()
        e.weight_by_strength(strength)
    if inject > 0:
        e.inject_initial_scale(inject, inject * 1.1)
    e.set_transform(T0.A, T0.t, T0.c)
    if not sym:
        e.set_number_points(nm)  # forward classes: use every moving point
    out = dict(kind=kind, dof=dof, seed=seed, normalize=normalize, strength=strength, inject=inject,
               iterations=iters, A0=T0.A, t0=T0.t, c0=T0.c)
    out.update(fs_arrays("moving", pair.moving))
    out.update(fs_arrays("fixed", pair.fixed))
    for it in range(iters):
        conv = e.one_iteration(it)
        out[f"it{it}_converged"] = conv
        for d in ((0, 1) if sym else (0,)):
            f, t, w = e.matches(d)
            out[f"it{it}_d{d}_from"], out[f"it{it}_d{d}_to"], out[f"it{it}_d{d}_w"] = f, t, w
            A, tt, c, cov = e.get_transform(d)
            out[f"it{it}_d{d}_A"], out[f"it{it}_d{d}_t"], out[f"it{it}_d{d}_c"], out[f"it{it}_d{d}_cov"] = A, tt, c, cov
            out[f"it{it}_d{d}_scale"] = e.last_scale(d)
            out[f"it{it}_d{d}_werr"] = e.weighted_error(d)
        AtA, Atb = e.normal_equations()
        out[f"it{it}_AtA"], out[f"it{it}_Atb"] = AtA, Atb
        out[f"it{it}_rms"] = e.RMS_error()
        out[f"it{it}_number_points"] = e.number_points(0)
    return out


def main():
    for name in CASES:
        out = run_case(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
