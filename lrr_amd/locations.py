"""Config 5: many independent query locations against one volume pair, sharded over GPUs.

Mirrors the outer loop of ``rrl/itkLocationRegistration.cxx:2134-2138`` for the estimation part
only: for every query location the features inside the 50 mm ROIs become the active sets of ONE
re-used ``rrl_estimation_symmetric_ICP_matching_all<3,12>`` object (``set_features`` +
``set_transform``, ``itkLocationRegistration.cxx:1635-1637``), which is iterated until it reports
convergence or 70 iterations (``:2418``).  Ranks take static blocks of the locations
(``multi_gpu.shard``) and one all_gather returns every location's result to every rank.
"""
from __future__ import annotations

import numpy as np

from . import estimation as E
from . import multi_gpu, synth

MAX_ITERATIONS = 70  # rrl/itkLocationRegistration.cxx:2413-2418


def register_locations(pair: synth.VolumePair, rois, inits, rank: int = 0, world: int = 1,
                       seed: int = 7, gather_device=None):
    """rois: list of (centre_fixed, centre_moving, moving_idx, fixed_idx) (synth.query_rois);
    inits: list of synth.Transform, the initial transform of every location.
    Returns (rows of multi_gpu.RESULT_WIDTH for ALL locations, matched pairs processed by this rank)."""
    est = None
    rows, pairs = [], 0
    for i in multi_gpu.shard(len(rois), rank, world):
        _, _, mi, fi = rois[i]
        m_in, f_in = pair.moving.subset(mi), pair.fixed.subset(fi)
        T0 = inits[i]
        if est is None:
            est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c,
                              moving_inside=m_in, fixed_inside=f_in, seed=seed)
        else:
            est.set_features(m_in, f_in)
            est.set_transform(T0.A, T0.t, T0.c)
        converged, it = False, 0
        while not converged and it < MAX_ITERATIONS:
            converged = est.one_iteration(it)
            pairs += est.number_points(0) + est.number_points(1)
            it += 1
        A, t, c, _ = est.get_transform()
        rows.append(multi_gpu.pack_result(i, converged, it, A, t, c, est.weighted_error(), est.RMS_error()))
    if est is not None:
        est.close()
    local = np.stack(rows) if rows else np.zeros((0, multi_gpu.RESULT_WIDTH))
    return multi_gpu.gather_results(local, len(rois), device=gather_device), pairs
