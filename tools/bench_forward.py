"""Finest-level iteration time of the forward estimators (cdcl_/rrl_estimation_ICP_matching_all) on 200k x 200k features: a side measurement, not the BASELINE metric (bench.py)."""
import sys, time, json, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lrr_amd import estimation as E, synth
import torch
E.set_device(0)
n = 200000
for kind, dof, cls in (("affine", 12, E.CDCL_FORWARD), ("rigid", 6, E.CDCL_FORWARD), ("affine", 12, E.RRL_FORWARD)):
    pair = synth.make_pair(n, n, 20243, kind=kind)
    T0 = synth.perturbed(pair.planted, 3)
    est = E.Estimator(cls, dof, pair.moving, pair.fixed, T0.A, T0.t, T0.c, seed=7)
    est.normalize_matches()
    est.weight_by_strength(True)
    est.set_number_points(n)
    timer = E.DeviceTimer()
    for it in range(8):
        est.one_iteration(it)
    timer.synchronize()
    t0 = time.perf_counter()
    K = 100
    est.iterate(8, K)
    timer.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"class": cls, "dof": dof, "ms_per_iteration": 1e3 * dt / K, "pairs_per_s": n * K / dt}))
    est.close()
