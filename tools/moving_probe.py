"""Wall time per iteration of the moving-transform regime: set_transform(planted (+) perturbation) at the finest level, then
one_iteration until converged (a synchronisation per iteration: for the timeline tool tools/moving_trace.sh)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from lrr_amd import estimation as E, synth
E.hip_lib(); E.set_device(0)
n = 200000
pair = synth.make_pair(n, n, 20244, kind="affine")
T0 = synth.perturbed(pair.planted, 3)
est = E.Estimator(E.RRL_SYMMETRIC, 12, pair.moving, pair.fixed, T0.A, T0.t, T0.c, moving_inside=pair.moving, fixed_inside=pair.fixed, seed=7)
timer = E.DeviceTimer()
for r in range(3):
    Tp = synth.perturbed(pair.planted, 500 + r)
    est.set_transform(Tp.A, Tp.t, Tp.c)
    est.set_number_points(10**9)
    ts = []
    for it in range(14):
        timer.synchronize(); t0 = time.perf_counter()
        conv = est.one_iteration(it)
        timer.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
        if conv: break
    print(r, 'iterations', len(ts), 'ms per iteration', [round(x, 3) for x in ts], 'total', round(sum(ts), 3))
