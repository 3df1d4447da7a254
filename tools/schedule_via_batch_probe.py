"""Probe: the caller's faithful 1000 -> N run() of ONE 200k x 200k problem through the batch engine (one location whose ROIs are the
whole volumes: the device-side loop, list parts) beside the estimator class's run() — wall time of each, iterations, end transforms."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from lrr_amd import estimation as E  # noqa: E402
from lrr_amd import locations, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
pair = synth.make_pair(n, n, 20245, kind="affine")
T0 = synth.perturbed(pair.planted, 11)
timer = E.DeviceTimer()
big = 1e9
box = np.array([[-big, -big, -big, big, big, big]])
batch = E.LocationBatch(pair.moving, pair.fixed)
C = E.LocationBatch.COL
for rep in range(4):
    timer.synchronize()
    t0 = time.perf_counter()
    rows = batch.run(T0.A[None], T0.t[None], T0.c[None], box, box, np.array([7], dtype=np.uint64), max_iterations=70, diagnostics=False, check_valid=False)
    timer.synchronize()
    dt = time.perf_counter() - t0
    print("batch of one: %.3f ms, iterations %d converged %d pairs %d  phases %s" % (1e3 * dt, rows[0, C["iterations"]], rows[0, C["converged"]],
                                                                                      rows[0, C["matched_pairs"]], batch.last_timing()))
Ab, tb = rows[0, C["A"]].reshape(3, 3), rows[0, C["t"]]
for rep in range(4):
    est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7)
    est.set_subset_seed(locations.location_seed(7, 0)) if hasattr(locations, "location_seed") else None
    timer.synchronize()
    t0 = time.perf_counter()
    conv, its, pairs = est.run_counted(70)
    timer.synchronize()
    dt = time.perf_counter() - t0
    A, t, c, _ = est.get_transform(0)
    print("estimator run(): %.3f ms, iterations %d converged %d pairs %d" % (1e3 * dt, its, conv, pairs))
    est.close()
print("end transforms differ by", float(np.abs(A - Ab).max()), float(np.abs(t - tb).max()))
